"""pyrecode_amd - MI355X-native implementation of pyReCoDe's per-frame reduce -> bit-pack -> compress hot path
(and the reader's sparse expand), behind the reference's ReCoDeWriter / ReCoDeReader / merge_parts API.
Compute lives in librecode_hip.so (hand-written HIP for gfx950, C ABI in include/recode_hip.h)."""
__version__ = "0.1.0"
