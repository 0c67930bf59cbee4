// hip_emul.h -- TEST INFRASTRUCTURE ONLY: a minimal CPU stand-in for the HIP runtime and the gfx950
// device builtins that tts_amd/csrc uses, so that the UNMODIFIED kernel and host sources can be
// compiled with g++ -fsanitize=address,undefined and run on tiny inputs under the sanitizers
// (GPU AddressSanitizer is not available on the MI355X pool).  One OS thread per work-item, one
// block at a time; __syncthreads and the wave64 cross-lane operations are real barriers/exchanges.
// Nothing in tts_amd/ includes, builds or loads this: the product has no CPU path.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <sanitizer/asan_interface.h>

// ---- runtime ------------------------------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef struct emul_stream *hipStream_t;
typedef struct emul_event { std::chrono::steady_clock::time_point t; } *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamNonBlocking = 1 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "emulated error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
typedef void *hipDeviceptr_t;
static inline hipError_t hipMemsetD32Async(hipDeviceptr_t d, int v, size_t count, hipStream_t) { for (size_t i = 0; i < count; ++i) static_cast<int *>(d)[i] = v; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)malloc(1); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new emul_event(); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }   // launches run to completion in order
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
struct hipDeviceProp_t { int multiProcessorCount; };
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { p->multiProcessorCount = 2; return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
enum { hipHostMallocDefault = 0, hipEventDisableTiming = 2, hipErrorInvalidDevice = 101 };
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new emul_event(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
#define HIP_SYMBOL(x) (&(x))
static inline hipError_t hipGetSymbolAddress(void **p, const void *sym) { *p = const_cast<void *>(sym); return hipSuccess; }   // a "__device__" array is host memory here

// ---- kernel language ----------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static           /* static LDS arrays: one block runs at a time */
#define __align__(n) __attribute__((aligned(n)))
#define __HIP_MEMORY_SCOPE_WORKGROUP 2

struct dim3 {
    uint32_t x, y, z;
    dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { uint32_t x, y, z, w; } __attribute__((aligned(16)));

namespace emul {
constexpr size_t LDS_MAX = 160 * 1024;
constexpr uint32_t WAVE = 64;

struct Barrier {                     // counts only threads that are still running the kernel
    std::mutex m;
    std::condition_variable cv;
    uint32_t expected = 0, waiting = 0;
    uint64_t gen = 0;
    void reset(uint32_t n) { expected = n; waiting = 0; }
    void wait()
    {
        std::unique_lock<std::mutex> lk(m);
        const uint64_t g = gen;
        if (++waiting >= expected) { waiting = 0; ++gen; cv.notify_all(); return; }
        cv.wait(lk, [&] { return gen != g; });
    }
    void drop()                      // a thread left the kernel: later barriers no longer wait for it
    {
        std::unique_lock<std::mutex> lk(m);
        if (expected) --expected;
        if (expected && waiting >= expected) { waiting = 0; ++gen; cv.notify_all(); }
    }
};

struct WaveState {
    Barrier bar;
    uint64_t slot[WAVE];
};

struct Block {
    Barrier bar;
    std::vector<WaveState> waves;
};

extern Block *g_block;
extern unsigned char g_lds[LDS_MAX] __attribute__((aligned(256)));
extern thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;

static inline void *dyn_lds() { return g_lds; }

void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()> &body);

static inline WaveState &my_wave() { return g_block->waves[t_threadIdx.x / WAVE]; }

// every lane of the wave that is still running must make the same call (convergent, as on the GPU)
static inline uint64_t wave_exchange(uint64_t mine, uint32_t from_lane, bool *valid = nullptr)
{
    WaveState &w = my_wave();
    const uint32_t lane = t_threadIdx.x % WAVE;
    w.slot[lane] = mine;
    w.bar.wait();
    const uint64_t v = w.slot[from_lane % WAVE];
    if (valid) *valid = from_lane < WAVE;
    w.bar.wait();
    return v;
}
}   // namespace emul

#define threadIdx emul::t_threadIdx
#define blockIdx emul::t_blockIdx
#define blockDim emul::t_blockDim
#define gridDim emul::t_gridDim

static inline void __syncthreads() { emul::g_block->bar.wait(); }

#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    emul::launch((grid), (block), (lds), [&]() { kernel(__VA_ARGS__); })

template <typename T>
static inline T __shfl_down(T v, unsigned off, int = 64)
{
    const uint32_t lane = emul::t_threadIdx.x % emul::WAVE;
    bool ok;
    uint64_t r = emul::wave_exchange((uint64_t)v, lane + off < emul::WAVE ? lane + off : lane, &ok);
    return (T)r;
}
template <typename T>
static inline T __shfl(T v, int src_lane, int = 64)
{
    return (T)emul::wave_exchange((uint64_t)v, (uint32_t)src_lane & 63u);
}
static inline uint64_t __ballot(bool pred)
{
    emul::WaveState &w = emul::my_wave();
    const uint32_t lane = emul::t_threadIdx.x % emul::WAVE;
    const uint32_t base = (emul::t_threadIdx.x / emul::WAVE) * emul::WAVE;
    w.slot[lane] = pred ? 1u : 0u;
    w.bar.wait();
    uint64_t m = 0;
    for (uint32_t l = 0; l < emul::WAVE && base + l < emul::t_blockDim.x; ++l) m |= (uint64_t)(w.slot[l] & 1u) << l;
    w.bar.wait();
    return m;
}
static inline uint64_t wall_clock64() { return 0; }
static inline int __popcll(uint64_t x) { return __builtin_popcountll(x); }
static inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_RELAXED)
#define __hip_atomic_fetch_or(p, v, order, scope) __atomic_fetch_or((p), (v), __ATOMIC_RELAXED)
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), __ATOMIC_RELAXED)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), __ATOMIC_RELAXED)
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline int __builtin_amdgcn_readfirstlane(int x)    // every live lane of the wave calls it
{
    return (int)(uint32_t)emul::wave_exchange((uint64_t)(uint32_t)x, 0u);
}
static inline uint32_t __builtin_amdgcn_bitop3_b32(uint32_t a, uint32_t b, uint32_t c, uint32_t tt)
{
    uint32_t r = 0;
    for (uint32_t i = 0; i < 8; ++i)
        if ((tt >> i) & 1u) r |= ((i & 4u) ? a : ~a) & ((i & 2u) ? b : ~b) & ((i & 1u) ? c : ~c);
    return r;
}
