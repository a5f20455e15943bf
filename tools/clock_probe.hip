// Shader clock under load: one wave spins for ~20 ms of wall clock and reports s_memtime (shader cycles) against
// s_memrealtime (100 MHz) - run it beside `bench.py` to see what frequency the VALU-bound stages really get.
// Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/clock_probe tools/clock_probe.hip ; usage: clock_probe [samples] [ms]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <time.h>

__global__ void k_probe(unsigned long long* out, unsigned long long ticks)
{
    unsigned long long c0 = clock64(), w0 = wall_clock64(), w = w0;
    unsigned v = threadIdx.x, n = 0;
    while (w - w0 < ticks) {
#pragma unroll
        for (int i = 0; i < 256; i++) asm volatile("v_pk_add_i16 %0, %0, %0" : "+v"(v));
        n += 256;
        w = wall_clock64();
    }
    unsigned long long c1 = clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w - w0; out[2] = n; out[3] = v; }
}

int main(int argc, char** argv)
{
    int samples = argc > 1 ? atoi(argv[1]) : 20, ms = argc > 2 ? atoi(argv[2]) : 20;
    unsigned long long* d; unsigned long long h[4];
    hipMalloc(&d, 32);
    for (int s = 0; s < samples; s++) {
        k_probe<<<1, 64>>>(d, (unsigned long long)ms * 100000ull);
        hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        double sec = h[1] / 1e8;
        { struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts); printf("%ld.%03ld ", (long)ts.tv_sec, ts.tv_nsec / 1000000); }
        printf("sample %2d: shader clock %.0f MHz; this wave issued %.1f M dependent packed VALU ops/s (cycles each: %.2f)\n", s, h[0] / sec / 1e6,
               (double)h[2] / sec / 1e6, h[0] / (double)h[2]);
        fflush(stdout);
        usleep(200000);
    }
    return 0;
}
