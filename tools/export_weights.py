"""Export the reference's prod_net weights to the blob the native library loads.

Run in the build container only (reads /root/reference); the output files are
committed under cut-detection_amd/frameID/prod_net/ and travel with the repo.
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cut-detection_amd"))
from cutdet import weights  # noqa: E402

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/frameID/prod_net"
OUT = os.path.join(ROOT, "cut-detection_amd", "frameID", "prod_net")

blob, params = weights.pack_reference_files(
    os.path.join(REF, "init_model_model_params.json"),
    os.path.join(REF, "init_model_classifier_conv.pt"),
    os.path.join(REF, "init_model_classifier_linear.pt"))
with open(os.path.join(OUT, "prod_net_f32.bin"), "wb") as f:
    f.write(blob)
manifest = {
    "model_params": params,
    "blob": "prod_net_f32.bin",
    "sha256": hashlib.sha256(blob).hexdigest(),
    "tensors": [{"name": n, "shape": list(s)} for n, s in weights.tensor_plan()],
}
with open(os.path.join(OUT, "prod_net_manifest.json"), "w") as f:
    json.dump(manifest, f, indent=1)
print(len(blob), manifest["sha256"])
