// SALU issue ceiling of one MI355X (gfx950): how many scalar instructions per second does the chip issue when every CU's scalar unit is
// kept busy?  bench.py prices the wave engine's scalar pipe against 256 CUs x 2.4 GHz x 1 instruction per clock = 614.4 G/s; the
// micro-architecture guide gives no scalar ceiling, so this measures it.
//   hipcc --offload-arch=gfx950 -O3 scripts/salu_microbench.hip -o build/salu_microbench && build/salu_microbench
// Each wave runs ITERS x 64 s_add_u32 / s_xor_b32 in NCHAIN independent dependency chains; waves per CU is swept (4 = one per SIMD ... 32).
// Output: one JSON line per (waves per CU, chains) with G instr/s and instructions per clock and CU at the measured shader clock.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                            \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));     \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

// 64 scalar ALU instructions per loop body, in NCHAIN independent chains (1, 2 or 4), plus s_sub / s_cmp / s_cbranch of the loop (3 more,
// counted).  The asm is volatile and its result is written at the end, so nothing is folded.
template <int NCHAIN>
__global__ void k_salu(uint32_t* out, uint32_t iters, unsigned long long* clocks) {
    uint32_t a = blockIdx.x, b = blockIdx.x + 1, c = blockIdx.x + 2, d = blockIdx.x + 3;
    const unsigned long long t0 = __builtin_readcyclecounter();
    // ONE asm statement per loop body (the compiler pads separate statements with s_nop) and SCC declared clobbered: s_add_u32
    // writes it, and without the clobber the loop's own s_cmp was scheduled ahead of the body (a loop that never ends).
#define R2(x) x x
#define R4(x) R2(R2(x))
#define R8(x) R2(R4(x))
#define R16(x) R2(R8(x))
#define R32(x) R2(R16(x))
    for (uint32_t it = 0; it < iters; ++it) {
        if (NCHAIN == 1) {
            asm volatile(R32("s_add_u32 %0, %0, 0x11\n s_xor_b32 %0, %0, 0x5a\n") : "+s"(a) : : "scc");
        } else if (NCHAIN == 2) {
            asm volatile(R16("s_add_u32 %0, %0, 0x11\n s_add_u32 %1, %1, 0x13\n s_xor_b32 %0, %0, 0x5a\n s_xor_b32 %1, %1, 0x3c\n") : "+s"(a), "+s"(b) : : "scc");
        } else {
            asm volatile(R8("s_add_u32 %0, %0, 0x11\n s_add_u32 %1, %1, 0x13\n s_add_u32 %2, %2, 0x17\n s_add_u32 %3, %3, 0x19\n"
                            "s_xor_b32 %0, %0, 0x5a\n s_xor_b32 %1, %1, 0x3c\n s_xor_b32 %2, %2, 0x66\n s_xor_b32 %3, %3, 0x71\n")
                         : "+s"(a), "+s"(b), "+s"(c), "+s"(d)
                         :
                         : "scc");
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) {
        out[blockIdx.x] = a ^ b ^ c ^ d;
        if (blockIdx.x == 0) *clocks = t1 - t0;
    }
}

template <int NCHAIN>
static void run(int waves_per_cu, int n_cu, uint32_t iters, uint32_t* d_out, unsigned long long* d_clk) {
    // one wave per workgroup: the dispatcher spreads workgroups over the CUs; grid = waves_per_cu x CUs is exactly one residency
    const int grid = waves_per_cu * n_cu;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_salu<NCHAIN>, dim3(grid), dim3(64), 0, 0, d_out, iters / 8, d_clk);  // warm
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_salu<NCHAIN>, dim3(grid), dim3(64), 0, 0, d_out, iters, d_clk);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long clk = 0;
    CHK(hipMemcpy(&clk, d_clk, sizeof(clk), hipMemcpyDeviceToHost));
    const double instr = (double)grid * (double)iters * 67.0;  // 64 + the loop's three
    const double gips = instr / (ms * 1e-3) / 1e9;
    // s_memtime ticks at a constant 100 MHz on this part; the shader clock is derived from the single-chain run instead (see main)
    printf("{\"waves_per_cu\": %d, \"chains\": %d, \"grid\": %d, \"ms\": %.3f, \"salu_ginstr_per_s\": %.1f, \"instr_per_ns_per_cu\": %.3f, \"frac_of_614.4\": %.3f}\n",
           waves_per_cu, NCHAIN, grid, ms, gips, gips / n_cu, gips / 614.4);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", prop.name, n_cu, prop.clockRate / 1000);
    uint32_t* d_out;
    unsigned long long* d_clk;
    CHK(hipMalloc(&d_out, 64 * 1024 * sizeof(uint32_t)));
    CHK(hipMalloc(&d_clk, sizeof(unsigned long long)));
    const uint32_t iters = 200000;
    for (int w : {1, 4, 8, 16, 24, 32}) {
        run<1>(w, n_cu, iters, d_out, d_clk);
        run<2>(w, n_cu, iters, d_out, d_clk);
        run<4>(w, n_cu, iters, d_out, d_clk);
    }
    return 0;
}
