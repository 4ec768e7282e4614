"""libspm_amd -- MI355X-native online pattern matching behind libspm's matcher API.

Only what the hot path needs lives here: csrc/ (HIP kernels + the C ABI of include/spm_hip.h), the ctypes binding
(capi), a thin object layer (engine) and the multi-GPU hit gather (dist).  The C++ mirror of the reference's
header-only API is in include/libspm/.
"""
from . import capi  # noqa: F401
from .capi import (ALGO_HORSPOOL, ALGO_MYERS, ALGO_MYERS_PREFIX, ALGO_SHIFTOR, ENGINE_AUTO, ENGINE_BRUTE,  # noqa: F401
                   ENGINE_FILTER, SCAN_DEFER, SCAN_IGNORE_PACKED, SpmError)
from .engine import (ALLELE_DTYPE, HIT_DTYPE, JST_HIT_DTYPE, Context, Hits, Jst, JstHits, PatternSet, Text, scan, scan_segments,
                     synth_variants,  # noqa: F401
                     synth_pattern, synth_repeat_pattern, synth_repeat_text)
