#!/bin/bash
# A/B on the range call, interleaved so that box-to-box differences and drift cancel:
#   tools/ab_range.sh <variant> <rounds> [range_stages args]
# <variant> is a build name (default build vs lib/variants/libcutdet_<name>.so) or an environment setting
# NAME=VALUE (same build, with and without it).
V=$1; R=$2; shift 2
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
if [[ "$V" == *=* ]]; then SET="$V"; else
  SET="CUTDET_LIB=$PWD/cut-detection_amd/lib/variants/libcutdet_$V.so"
  [ -f "${SET#CUTDET_LIB=}" ] || { echo "no such variant build: ${SET#CUTDET_LIB=}"; exit 1; }
fi
for r in $(seq 1 $R); do
  echo "default: $(timeout -k 10 300 python3 -u tools/range_stages.py "$@" | grep 'range entry')" || exit 1
  echo "$V: $(env "$SET" timeout -k 10 300 python3 -u tools/range_stages.py "$@" | grep 'range entry')" || exit 1
done
