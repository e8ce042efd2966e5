"""Import BEFORE m17hip: the measurement build of the library (make -C m17-cxx-demod_amd/csrc tools -> libm17hip_tools.so: section timers,
per-wave times, schedule experiment knobs) unless M17HIP_LIB names another build."""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("M17HIP_LIB", os.path.join(ROOT, "m17-cxx-demod_amd", "libm17hip_tools.so"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
