# bf16-mode stage times for each library variant.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for lib in cut-detection_amd/lib/libcutdet.so $(ls cut-detection_amd/lib/variants/*.so 2>/dev/null); do
  echo "== $lib"
  CUTDET_LIB=$PWD/$lib timeout -k 10 200 python tools/check_bf16.py 2>gpurun_out/bv.err | tail -2 || { tail -5 gpurun_out/bv.err; exit 1; }
done
