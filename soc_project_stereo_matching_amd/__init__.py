"""soc_project_stereo_matching_amd -- MI355X-native Semi-Global Matching behind the reference's C boundary.

The product is ``libsgm_mi355x.so`` (C host + hand-written gfx950 HIP kernels, see ``csrc/``) whose
entry points are the reference's ``SGM_Initialize / SGM_Reset / SGM_Match``
(reference: SemiGlobalMatching/SemiGlobalMatching/SemiGlobalMatching.h:78-80).  This Python package
is only a ctypes mirror of that C interface for tests and benchmarks; it contains no compute and
has no CPU fallback -- importing works anywhere, calling into the library needs a gfx950 GPU.
"""
from .sgm import (SGM, SGMInstance, SGMOption, default_option, library_path, load_library,  # noqa: F401
                  STAGE_NAMES, synth_pair)
from .sharding import SGMStream, frames_of_rank, match_sharded  # noqa: F401

__all__ = ["SGM", "SGMInstance", "SGMOption", "default_option", "library_path", "load_library", "STAGE_NAMES",
           "synth_pair", "SGMStream", "frames_of_rank", "match_sharded"]
