"""fast-livo_amd: MI355X-native ESKF hot path for FAST-LIVO (residual/Jacobian assembly + iterated
error-state Kalman update).  Product code lives in csrc/ (HIP kernels + C ABI, built into
libfastlivo_hip.so); `capi` is the ctypes binding used by tests and bench, `synth` the seeded
synthetic-frame generator of SURVEY.md section 8(d).  Nothing here imports oracle/."""
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
LIB_PATH = os.environ.get("FL_LIB_PATH") or os.path.join(PKG_DIR, "libfastlivo_hip.so")   # FL_LIB_PATH: A/B builds (tools/)
