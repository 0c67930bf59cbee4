// hip_emul.cpp -- the launch loop of the CPU stand-in (see hip_emul.h).  TEST INFRASTRUCTURE ONLY.
#include "hip_emul.h"

#include <stdio.h>

namespace emul {
Block *g_block = nullptr;
unsigned char g_lds[LDS_MAX] __attribute__((aligned(256)));
thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;

void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()> &body)
{
    if (block.y != 1 || block.z != 1 || grid.z != 1 || block.x == 0 || block.x > 1024 || lds_bytes > LDS_MAX) {
        fprintf(stderr, "emul::launch: unsupported launch shape\n");
        abort();
    }
    const uint32_t nthreads = block.x, nwaves = (nthreads + WAVE - 1) / WAVE;
    Block blk;
    blk.waves = std::vector<WaveState>(nwaves);
    g_block = &blk;
    Barrier between;                     // all threads, between blocks
    between.reset(nthreads);
    ASAN_UNPOISON_MEMORY_REGION(g_lds, LDS_MAX);
    memset(g_lds, 0xA5, LDS_MAX);        // LDS starts with garbage, as on the GPU
    ASAN_POISON_MEMORY_REGION(g_lds + lds_bytes, LDS_MAX - lds_bytes);

    auto worker = [&](uint32_t tid) {
        t_blockDim = block; t_gridDim = grid; t_threadIdx = dim3(tid, 0, 0);
        for (uint32_t by = 0; by < grid.y; ++by)
            for (uint32_t bx = 0; bx < grid.x; ++bx) {
                if (tid == 0) {
                    blk.bar.reset(nthreads);
                    for (uint32_t w = 0; w < nwaves; ++w)
                        blk.waves[w].bar.reset(std::min<uint32_t>(WAVE, nthreads - w * WAVE));
                }
                between.wait();
                t_blockIdx = dim3(bx, by, 0);
                body();
                blk.bar.drop();
                blk.waves[tid / WAVE].bar.drop();
                between.wait();
            }
    };
    std::vector<std::thread> th;
    th.reserve(nthreads);
    for (uint32_t t = 0; t < nthreads; ++t) th.emplace_back(worker, t);
    for (auto &t : th) t.join();
    ASAN_UNPOISON_MEMORY_REGION(g_lds, LDS_MAX);
    g_block = nullptr;
}
}   // namespace emul
