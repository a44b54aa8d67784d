// Shared host/device helpers for libbuffer_hip.so (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/buffer_hip.h"

#define WAVE 64

void buf_set_error(const char* fmt, ...);

#define BUF_CHECK_HIP(expr)                                                            \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) {                                                        \
            buf_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return BUF_EHIP;                                                           \
        }                                                                              \
    } while (0)

#define BUF_REQUIRE(cond, code, ...)                                                   \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            buf_set_error(__VA_ARGS__);                                                \
            return (code);                                                             \
        }                                                                              \
    } while (0)

#define BUF_LAUNCH_CHECK() BUF_CHECK_HIP(hipGetLastError())

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Bump allocator over a caller-provided device workspace.
struct WsCarver {
    char* base;
    size_t off, cap;
    bool ok;
    WsCarver(void* p, size_t bytes) : base((char*)p), off(0), cap(bytes), ok(true) {}
    template <typename T> T* take(size_t count)
    {
        off = align_up(off, 256);
        size_t bytes = count * sizeof(T);
        if (base && off + bytes > cap) ok = false;
        T* r = base ? (T*)(base + off) : nullptr;
        off += bytes;
        return r;
    }
    size_t used() const { return align_up(off, 256); }
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device only: the grant is remembered per
// (kernel, device), so a process that drives several GPUs raises the limit on each of them.
#define BUF_MAX_DEVICES 64
struct LdsGrant { size_t bytes[BUF_MAX_DEVICES]; };
int grant_dynamic_lds(const void* kernel, size_t bytes, LdsGrant& g);

// In-place exclusive scan of int32 data[n] on `stream`; tmp must hold scan_tmp_ints() ints.
size_t scan_tmp_ints();
int exclusive_scan_i32(int* data, long long n, int* tmp, int* total_out, hipStream_t stream);

// ---- device helpers ---------------------------------------------------------------------
// Hand-over of LDS data between the lanes of ONE wavefront: the scheduling barrier alone is no memory fence for the compiler
// (IntrNoMem), so it is paired with wavefront-scope release / acquire fences (no instructions on a single wave).
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Reference distance: result = 0; result += dx*dx; += dy*dy; += dz*dz  (nanoflann.hpp:433-441).
// __fmul_rn/__fadd_rn are never contracted into FMA.
__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz)
{
    float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
    float r = __fmul_rn(dx, dx);
    r = __fadd_rn(r, __fmul_rn(dy, dy));
    r = __fadd_rn(r, __fmul_rn(dz, dz));
    return r;
}

// XCD-aware block order (speed only; MI355X_MICROARCH "Workgroup dispatch": blocks b and b + 8 share an XCD and its L2).
// Logical block of physical block `bid`: the blocks of one XCD take a CONTIGUOUS range of logical blocks, so neighbouring
// work items -- which gather from the same rows -- meet in one L2 instead of being dealt over all eight.  A bijection on [0, nb).
__device__ __forceinline__ int xcd_contiguous_block(int bid, int nb)
{
    const int x = bid & 7, idx = bid >> 3, q = nb >> 3, r = nb & 7;
    return x * q + min(x, r) + idx;
}

// element b such that off[b] <= i < off[b+1]
__device__ __forceinline__ int find_elem(const int* __restrict__ off, int nb, int i)
{
    int lo = 0, hi = nb - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= i) lo = mid; else hi = mid - 1;
    }
    return lo;
}
