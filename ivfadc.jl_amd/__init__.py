"""ivfadc.jl_amd -- MI355X (gfx950) implementation of IVFADC.jl's knn_search hot path.

Exports mirror /root/reference/src/IVFADC.jl:16-20 (`push!` & co. lose the `!`).
The compute path is csrc/ (hand-written HIP behind the C ABI of include/ivfadc_hip.h);
this package is the thin host side and has no CPU fallback.
"""
from ._native import IVFADCError, build as build_library, lib as load_library, needs_build  # noqa: F401
from .index import (IVFADCIndex, InvertedList, NaiveQuantizer, CodeBook, knn_search, knn_search_batches, push, pushfirst, pop, popfirst,  # noqa: F401
                    delete_from_index, comm_unique_id)
from .persistency import save_ivfadc_index, load_ivfadc_index  # noqa: F401
from . import distributed, trainer  # noqa: F401

__all__ = ["IVFADCIndex", "delete_from_index", "knn_search", "save_ivfadc_index", "load_ivfadc_index",
           "push", "pushfirst", "pop", "popfirst"]
