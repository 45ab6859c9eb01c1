"""ctypes loader of vdjer_amd/libvdjx.so (the HIP hot path).  There is no CPU fallback: if the library is
missing or cannot be loaded this raises, loudly."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# VDJX_LIB_PATH: another build of the same library (profiles/: the -DVDJX_ABLATE build, `make -C vdjer_amd/csrc ablate`)
LIB_PATH = os.environ.get("VDJX_LIB_PATH") or os.path.join(_HERE, "libvdjx.so")
_lib = None

# every symbol include/vdjx.h declares
SYMBOLS = [
    "vdjx_last_error", "vdjx_version", "vdjx_init", "vdjx_shutdown", "vdjx_sync", "vdjx_trim", "vdjx_read_index_drop", "vdjx_device_copy",
    "vdjx_pool_load", "vdjx_pool_load_forward", "vdjx_pool_load_forward_begin", "vdjx_pool_wait", "vdjx_packed_read_bytes", "vdjx_pack_reads", "vdjx_pool_load_packed", "vdjx_pool_load_packed_begin", "vdjx_pool_load_device", "vdjx_pool_records", "vdjx_pool_free",
    "vdjx_anchor_sets_load", "vdjx_anchor_probe", "vdjx_index_generate", "vdjx_anchor_sets_from_anchors",
    "vdjx_kmer_build", "vdjx_graph_nodes", "vdjx_graph_pre_nodes", "vdjx_graph_export", "vdjx_graph_export_begin", "vdjx_graph_export_end", "vdjx_graph_block_layout", "vdjx_graph_export_block", "vdjx_graph_export_block_begin", "vdjx_graph_free",
    "vdjx_vregion_load", "vdjx_root_score", "vdjx_graph_roots", "vdjx_root_part", "vdjx_root_score_graph", "vdjx_root_score_graph_begin", "vdjx_root_score_graph_end",
    "vdjx_read_index_build", "vdjx_read_index_build_device", "vdjx_read_index_build_begin", "vdjx_read_index_build_device_begin", "vdjx_read_index_build_end", "vdjx_window_score", "vdjx_window_pairs", "vdjx_window_pairs_fetch", "vdjx_window_cover", "vdjx_map_emit", "vdjx_map_emit_begin", "vdjx_map_emit_end", "vdjx_sam_names_load", "vdjx_sam_text", "vdjx_sam_blocks", "vdjx_sam_merge", "vdjx_rows_scatter",
    "vdjx_host_alloc", "vdjx_host_free", "vdjx_host_take_rows",
    "vdjx_stat", "vdjx_profile_enable", "vdjx_profile_only", "vdjx_profile_reset", "vdjx_profile_count", "vdjx_profile_get",
    "vdjx_shard_begin", "vdjx_shard_begin_share", "vdjx_shard_free", "vdjx_shard_record_bytes", "vdjx_shard_count", "vdjx_shard_geometry", "vdjx_shard_symmetric", "vdjx_shard_geometry2", "vdjx_shard_local", "vdjx_shard_local_fill",
    "vdjx_shard_merge", "vdjx_shard_queries", "vdjx_shard_reply", "vdjx_shard_resolve",
    "vdjx_shard_survivors", "vdjx_shard_edges", "vdjx_shard_finish",
]


class VdjxError(RuntimeError):
    pass


class CovParams(C.Structure):
    _fields_ = [("eval_start", C.c_int), ("eval_stop", C.c_int), ("read_span", C.c_int), ("mate_span", C.c_int),
                ("insert_low", C.c_int), ("insert_high", C.c_int), ("floor", C.c_int)]


class Pair(C.Structure):
    _fields_ = [("pair_id", C.c_uint32), ("rec1", C.c_uint32), ("rec2", C.c_uint32),
                ("pos1", C.c_int16), ("pos2", C.c_int16), ("insert", C.c_int16),
                ("rc1", C.c_uint8), ("rc2", C.c_uint8)]


def build() -> str:
    """Compile libvdjx.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc")])
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64; whichever HIP runtime is loaded first serves the whole process.  When torch
    # is used in the same process (bench.py, shard.py) it must come first so that device pointers are shared.
    try:
        import torch  # noqa: F401
    except Exception:  # torch is optional for the single-GPU C ABI
        pass
    if not os.path.exists(LIB_PATH):
        raise VdjxError(f"{LIB_PATH} is missing: build it with `make -C vdjer_amd/csrc` "
                        "(there is no CPU fallback for the hot path)")
    L = C.CDLL(LIB_PATH)
    vp, sz, i32, u32 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint32
    L.vdjx_last_error.restype = C.c_char_p
    L.vdjx_version.restype = C.c_char_p
    L.vdjx_init.argtypes = [i32, C.POINTER(vp)]
    L.vdjx_shutdown.argtypes = [vp]
    L.vdjx_shutdown.restype = None
    L.vdjx_sync.argtypes = [vp]
    L.vdjx_pool_load.argtypes = [vp, vp, sz, vp, sz, i32, C.POINTER(vp)]
    L.vdjx_pool_load_device.argtypes = [vp, vp, sz, vp, sz, i32, C.POINTER(vp)]
    L.vdjx_pool_load_forward.argtypes = [vp, vp, sz, vp, sz, i32, C.POINTER(vp)]
    L.vdjx_pool_load_forward_begin.argtypes = [vp, vp, sz, vp, sz, i32, C.POINTER(vp)]
    L.vdjx_pool_load_packed.argtypes = [vp, vp, sz, vp, sz, i32, C.POINTER(vp)]
    L.vdjx_pool_load_packed_begin.argtypes = [vp, vp, sz, vp, sz, i32, C.POINTER(vp)]
    L.vdjx_packed_read_bytes.argtypes = [i32]
    L.vdjx_packed_read_bytes.restype = C.c_size_t
    L.vdjx_pack_reads.argtypes = [vp, sz, i32, vp]
    L.vdjx_pool_wait.argtypes = [vp]
    L.vdjx_pool_records.argtypes = [vp]
    L.vdjx_pool_records.restype = sz
    L.vdjx_pool_free.argtypes = [vp]
    L.vdjx_pool_free.restype = None
    L.vdjx_anchor_sets_load.argtypes = [vp, vp, sz, vp, sz]
    L.vdjx_anchor_probe.argtypes = [vp, C.c_char_p, i32, vp, vp]
    L.vdjx_kmer_build.argtypes = [vp, vp, i32, i32, i32, C.POINTER(vp)]
    L.vdjx_graph_nodes.argtypes = [vp]
    L.vdjx_graph_nodes.restype = sz
    L.vdjx_graph_pre_nodes.argtypes = [vp]
    L.vdjx_graph_pre_nodes.restype = sz
    L.vdjx_graph_export.argtypes = [vp] + [vp] * 10
    L.vdjx_graph_export_begin.argtypes = [vp] + [vp] * 10
    L.vdjx_graph_export_end.argtypes = [vp]
    L.vdjx_graph_block_layout.argtypes = [vp, vp, vp]
    L.vdjx_graph_export_block.argtypes = [vp, vp]
    L.vdjx_graph_export_block_begin.argtypes = [vp, vp]
    L.vdjx_graph_free.argtypes = [vp]
    L.vdjx_graph_free.restype = None
    L.vdjx_vregion_load.argtypes = [vp, C.POINTER(C.c_char_p), sz, i32]
    L.vdjx_root_score.argtypes = [vp, C.c_char_p, sz, i32, i32, vp]
    L.vdjx_read_index_build.argtypes = [vp, vp, vp, vp, vp, vp, u32]
    L.vdjx_read_index_build_device.argtypes = [vp, vp, vp, vp, vp, vp, u32]
    L.vdjx_read_index_build_begin.argtypes = [vp, vp, vp, vp, vp, vp, u32]
    L.vdjx_read_index_build_device_begin.argtypes = [vp, vp, vp, vp, vp, vp, u32]
    L.vdjx_read_index_build_end.argtypes = [vp]
    L.vdjx_window_score.argtypes = [vp, C.c_char_p, sz, i32, C.POINTER(CovParams), vp, vp]
    L.vdjx_map_emit.argtypes = [vp, C.c_char_p, sz, i32, vp, vp]
    L.vdjx_window_pairs.argtypes = [vp, C.c_char_p, sz, i32, vp, vp]
    L.vdjx_window_pairs_fetch.argtypes = [vp, vp, sz, vp]
    L.vdjx_window_cover.argtypes = [vp, sz, i32, i32, C.POINTER(CovParams), vp, sz, vp, vp]
    L.vdjx_map_emit_begin.argtypes = [vp, C.c_char_p, sz, i32, vp, vp]
    L.vdjx_map_emit_end.argtypes = [vp]
    L.vdjx_sam_names_load.argtypes = [vp, C.c_char_p, vp, u32]
    L.vdjx_sam_text.argtypes = [vp, C.c_char_p, sz, i32, C.c_char_p, vp, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64)]
    L.vdjx_sam_blocks.argtypes = [vp, C.c_char_p, sz, i32, C.c_char_p, vp, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.vdjx_sam_merge.argtypes = [vp, C.c_uint64, C.c_uint64, vp, vp, vp, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64)]
    L.vdjx_trim.argtypes = [vp]
    L.vdjx_read_index_drop.argtypes = [vp]
    L.vdjx_device_copy.argtypes = [vp, vp, vp, C.c_size_t]
    L.vdjx_graph_roots.argtypes = [vp]
    L.vdjx_graph_roots.restype = C.c_size_t
    L.vdjx_root_part.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.vdjx_root_part.restype = C.c_size_t
    L.vdjx_root_score_graph.argtypes = [vp, vp, C.c_int, C.c_uint32, C.c_uint32, vp, vp]
    L.vdjx_root_score_graph_begin.argtypes = [vp, vp, C.c_int, C.c_uint32, C.c_uint32, vp, vp]
    L.vdjx_root_score_graph_end.argtypes = [vp]
    L.vdjx_index_generate.argtypes = [vp, vp, C.c_size_t, C.c_uint64, C.c_uint64, C.c_int, C.c_uint64, C.POINTER(C.c_uint64), vp, vp]
    L.vdjx_anchor_sets_from_anchors.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_int]
    L.vdjx_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.vdjx_host_free.argtypes = [vp, vp]
    L.vdjx_host_free.restype = None
    L.vdjx_host_take_rows.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, vp, C.c_size_t]
    L.vdjx_stat.argtypes = [vp, C.c_char_p]
    L.vdjx_stat.restype = C.c_uint64
    L.vdjx_profile_enable.argtypes = [vp, i32]
    L.vdjx_profile_only.argtypes = [vp, C.c_char_p]
    L.vdjx_profile_reset.argtypes = [vp]
    L.vdjx_profile_count.argtypes = [vp]
    L.vdjx_profile_get.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    _lib = L
    return L


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise VdjxError(f"{what or 'vdjx'} failed ({rc}): {lib().vdjx_last_error().decode(errors='replace')}")
