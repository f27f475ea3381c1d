"""oracle/ -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).

A CPU restatement (plain PyTorch fp32 on the host) of the hot path of
aanna0701/face-recognition-pytorch: IR-50-layout ResNet backbone
(nets/resnet.py), additive angular margin (nets/ArcFace.py), class-sharded
PartialFC head with distributed softmax cross-entropy (nets/PartialFC.py) and
the training step composition (model/FR_PartialFC.py:162-193).

Pinning: every function here is checked against golden vectors produced by
importing the real reference modules from /root/reference in the build
container (tools/make_golden.py -> tests/golden/*.npz; tests/test_oracle_*.py).
The reference itself has no tests or fixtures, so these
generated vectors are the only pin.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package.  The product (face-recognition-pytorch_amd/) never does;
it raises if the HIP library is missing.
"""
