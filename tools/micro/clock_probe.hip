// The shader clock the chip holds while other work runs: one wavefront stamps s_memtime (core clock) and s_memrealtime
// (100 MHz) around a spin of about `ms` milliseconds; clock = d(memtime) / d(memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS
// give-back, item 6).  Built as a small shared library, driven by tools/clock_under_load.py through ctypes:
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/micro/clock_probe.hip -o tools/micro/libclock_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void k_clock_probe(uint64_t *out, uint64_t real_ticks) {
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    uint64_t r1 = r0;
    while (r1 - r0 < real_ticks) {
        __builtin_amdgcn_s_sleep(32);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}

extern "C" int clock_probe_launch(void *stream, uint64_t *out_dev, double ms) {
    k_clock_probe<<<1, 64, 0, (hipStream_t)stream>>>(out_dev, (uint64_t)(ms * 1e-3 * 100e6));
    return (int)hipGetLastError();
}
