// Fill every CU's LDS with a word pattern (tools/lds_poison_probe.py): a workgroup's LDS is NOT cleared when it starts -- it holds what
// the workgroups before it (of any kernel, of any PROCESS) left there.  A kernel that reads an LDS word it never wrote computes with
// that garbage; on a device it has to itself the garbage is its own previous launch's (finite, stable) data and nothing shows.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/ubench/lds_poison.hip -o tools/ab/liblds_poison.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void __launch_bounds__(256) ldsPoisonKernel(uint32_t pattern, uint32_t words, unsigned long long ticks)
{
    extern __shared__ uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) lds[i] = pattern;
    __syncthreads();
    // stay for a while so that the grid's workgroups are spread over all CUs instead of a few CUs serving all of them
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (lds[(threadIdx.x * 37u) % words] != pattern) __builtin_trap();     // (keeps the stores)
}

extern "C" int lds_poison(uint32_t pattern, void *stream)
{
    static bool once = false;
    if (!once) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(ldsPoisonKernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
        once = true;
    }
    // one 160 KB workgroup per CU at a time: 512 of them, each holding its CU for 20 us, cover the 256 CUs twice
    hipLaunchKernelGGL(ldsPoisonKernel, dim3(512), dim3(256), 160 * 1024, reinterpret_cast<hipStream_t>(stream), pattern, 160u * 1024u / 4u, 2000ull);
    return int(hipGetLastError());
}

// what fresh workgroups find: the fraction of LDS words equal to `pattern` over 512 workgroups that only read (out[0] = matching words, out[1] = words read)
__global__ void __launch_bounds__(256) ldsPeekKernel(uint32_t pattern, uint32_t words, unsigned long long *out)
{
    extern __shared__ uint32_t lds[];
    unsigned long long n = 0;
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) n += lds[i] == pattern ? 1u : 0u;
    atomicAdd(out, n);
    if (threadIdx.x == 0) atomicAdd(out + 1, (unsigned long long)words);
}

extern "C" int lds_peek(uint32_t pattern, unsigned long long *d_out, void *stream)
{
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(ldsPeekKernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
    hipLaunchKernelGGL(ldsPeekKernel, dim3(512), dim3(256), 160 * 1024, reinterpret_cast<hipStream_t>(stream), pattern, 160u * 1024u / 4u, d_out);
    return int(hipGetLastError());
}
