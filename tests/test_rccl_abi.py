"""csrc/sgm_tiles.c binds RCCL at run time and declares the few functions it uses by hand (csrc/sgm_rccl_abi.h) -- among them
ncclCommInitRank, which takes the 128-byte ncclUniqueId BY VALUE.  No multi-rank communicator has met that code yet (one GPU per
box), so the declarations are pinned here against the real header of this ROCm: sizes, enum values and every parameter list."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

SRC = r'''
#include <rccl/rccl.h>
#include <type_traits>
#include "sgm_rccl_abi.h"

// the id: same size, byte alignment, trivially copyable standard layout -> passed in memory by the SysV x86-64 ABI either way
static_assert(sizeof(ncclUniqueId) == sizeof(rccl_uid) && sizeof(rccl_uid) == SGM_TILES_ID_BYTES && NCCL_UNIQUE_ID_BYTES == SGM_TILES_ID_BYTES, "id size");
static_assert(alignof(ncclUniqueId) == alignof(rccl_uid), "id alignment");
static_assert(std::is_trivially_copyable<ncclUniqueId>::value && std::is_standard_layout<ncclUniqueId>::value, "id class");
// handles, enums and results as the hand declarations pass them
static_assert(sizeof(ncclComm_t) == sizeof(void*) && std::is_pointer<ncclComm_t>::value, "ncclComm_t is a pointer");
static_assert(sizeof(hipStream_t) == sizeof(void*) && std::is_pointer<hipStream_t>::value, "hipStream_t is a pointer");
static_assert(sizeof(ncclDataType_t) == sizeof(int) && sizeof(ncclResult_t) == sizeof(int), "enums are ints");
static_assert((int)ncclUint8 == RCCL_UINT8 && (int)ncclSuccess == RCCL_SUCCESS, "enum values");
// parameter lists of the real prototypes
static_assert(std::is_same<decltype(&ncclGetUniqueId), ncclResult_t (*)(ncclUniqueId*)>::value, "ncclGetUniqueId");
static_assert(std::is_same<decltype(&ncclCommInitRank), ncclResult_t (*)(ncclComm_t*, int, ncclUniqueId, int)>::value, "ncclCommInitRank");
static_assert(std::is_same<decltype(&ncclCommDestroy), ncclResult_t (*)(ncclComm_t)>::value, "ncclCommDestroy");
static_assert(std::is_same<decltype(&ncclGroupStart), ncclResult_t (*)()>::value && std::is_same<decltype(&ncclGroupEnd), ncclResult_t (*)()>::value, "groups");
static_assert(std::is_same<decltype(&ncclSend), ncclResult_t (*)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)>::value, "ncclSend");
static_assert(std::is_same<decltype(&ncclRecv), ncclResult_t (*)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)>::value, "ncclRecv");
static_assert(std::is_same<decltype(&ncclGetErrorString), const char* (*)(ncclResult_t)>::value, "ncclGetErrorString");
// ... and the hand-declared pointer types, parameter for parameter
static_assert(std::is_same<decltype(rccl_api::CommInitRank), int (*)(void**, int, rccl_uid, int)>::value, "ours: CommInitRank");
static_assert(std::is_same<decltype(rccl_api::Send), int (*)(const void*, size_t, int, int, void*, void*)>::value, "ours: Send");
static_assert(std::is_same<decltype(rccl_api::Recv), int (*)(void*, size_t, int, int, void*, void*)>::value, "ours: Recv");
int main() { return 0; }
'''


def test_hand_declared_rccl_slice_matches_the_real_header(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists("/opt/rocm/include/rccl/rccl.h") or not os.path.exists(hipcc):
        pytest.skip("no rccl.h / hipcc on this machine")
    src = tmp_path / "abi.cpp"
    src.write_text(SRC)
    r = subprocess.run([hipcc, "-x", "hip", "--cuda-host-only", "-std=c++17", "-fsyntax-only", "-I/opt/rocm/include",
                        "-I", os.path.join(ROOT, "soc_project_stereo_matching_amd", "csrc"), str(src)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
