"""MI355X-native LDP denoising hot path (planner U-Net DDPM/DDIM loop, IDM loop, StableVAE encode) and its training step."""
import os as _os
import re as _re


def library_version_from_source() -> str:
    """The version string csrc/engine.hip compiles into ldp_version() ("ldp_hip X.Y.Z ..."): ONE number for the package and the library."""
    with open(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "csrc", "engine.hip")) as f:
        m = _re.search(r'"ldp_hip (\d+\.\d+\.\d+) ', f.read())
    return m.group(1) if m else "0"


__version__ = library_version_from_source()
