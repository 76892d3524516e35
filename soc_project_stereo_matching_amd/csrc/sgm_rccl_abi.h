/*
 * sgm_rccl_abi.h -- the slice of rccl.h that csrc/sgm_tiles.c binds at run time (dlopen / dlsym: the library neither links librccl
 * nor needs its header to build), declared by hand.  tests/test_rccl_abi.py compiles these declarations against the real
 * /opt/rocm/include/rccl/rccl.h and fails if a size, a value or a parameter list has drifted.
 */
#ifndef SGM_RCCL_ABI_H
#define SGM_RCCL_ABI_H

#include <stddef.h>
#include "../../include/sgm_tiles.h"

typedef struct { char internal[SGM_TILES_ID_BYTES]; } rccl_uid;      /* = ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128), passed BY VALUE */
typedef struct {
    void* lib;
    int (*GetUniqueId)(rccl_uid*);                                   /* ncclResult_t ncclGetUniqueId(ncclUniqueId*) */
    int (*CommInitRank)(void**, int, rccl_uid, int);                 /* ncclCommInitRank(ncclComm_t*, int nranks, ncclUniqueId, int rank) */
    int (*CommDestroy)(void*);                                       /* ncclCommDestroy(ncclComm_t) */
    int (*GroupStart)(void);
    int (*GroupEnd)(void);
    int (*Send)(const void*, size_t, int, int, void*, void*);        /* ncclSend(buf, count, ncclDataType_t, peer, ncclComm_t, hipStream_t) */
    int (*Recv)(void*, size_t, int, int, void*, void*);
    const char* (*GetErrorString)(int);
} rccl_api;
enum { RCCL_UINT8 = 1, RCCL_SUCCESS = 0 };                           /* ncclUint8, ncclSuccess */

#endif
