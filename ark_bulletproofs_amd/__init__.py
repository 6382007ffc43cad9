"""MI355X-native MSM + inner-product engine behind the hot path of FindoraNetwork/ark-bulletproofs.

The compute lives in libarkbp_hip.so (HIP kernels for gfx950 + C++ host logic, C ABI in
include/arkbp.h).  This package is the thin Python binding used by the tests and bench.py; names follow
the reference (BulletproofGens, PedersenGens, InnerProductProof, Prover, Verifier, batch_verify)."""
from .engine import Engine, SECQ256K1, ZORRO  # noqa: F401
from ._lib import ArkbpError, LIB_PATH  # noqa: F401
