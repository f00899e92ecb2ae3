"""supertonic_amd — MI355X-native Supertonic TTS inference engine (host-side Python mirror).

The compute path is the HIP library `libstn.so` (C-ABI in include/stn.h); importing the
package does not load it — `supertonic_amd.binding.load()` does, and fails loudly if the
library is missing (there is no CPU fallback)."""
from .arch import StnArch, default_arch, tiny_arch  # noqa: F401
