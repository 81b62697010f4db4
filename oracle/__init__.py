"""CPU oracle for the classify+segment hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, what the reference computes for the path
u8 frame -> resize/normalise -> prod_net logits -> argmax/max -> segments -> CSV.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
it; the product (cut-detection_amd/) never does and has no CPU fallback.

Pinning status
  net forward      pinned: golden logits produced by importing the reference
                   (tests/golden/make_golden.py -> tests/golden/net_golden.npz)
  segmentation     pinned: golden `te` dicts / CSV bytes produced by the reference
                   class (tests/golden/seg_golden.json)
  resize (P1)      PARITY UNPINNED: OpenCV is not installable here and the reference
                   holds no fixture for it; the 8U INTER_LINEAR algorithm is restated
                   from its published description.  The 1280x720 -> 256x144 case is a
                   pure point sample and does not depend on that restatement.
"""
