// Instruction-rate probes for the build kernels (not part of the product): what one SIMD sustains
// for the integer operations the k-mer hash is made of, and what LDS atomics cost at the bin
// counts the scatter / reduce kernels use.  Prints cycles per wave-instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench tools/ubench.hip && tools/ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kIters = 2048;   // loop trips
constexpr int kChains = 8;     // independent chains per trip

template <int OP>
__global__ __launch_bounds__(256) void valu_kernel(uint32_t *out, uint32_t seed)
{
    uint32_t a[kChains], b[kChains];
    uint64_t w[kChains];
    for (int i = 0; i < kChains; ++i) { a[i] = threadIdx.x * 2654435761u + i + seed; b[i] = a[i] ^ 0x9e3779b9u; w[i] = ((uint64_t)a[i] << 32) | b[i]; }
    const uint32_t K = 0x6659FD93u | (seed & 1u);
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < kChains; ++i) {
            if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(K));
            if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(K));
            if (OP == 3) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(K) : "vcc");
            if (OP == 4) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(w[i]));
            if (OP == 5) asm volatile("v_alignbit_b32 %0, %0, %1, 30" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 6) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 7) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(a[i]));
            if (OP == 8) asm volatile("v_ffbh_u32 %0, %0" : "+v"(a[i]));
            if (OP == 9) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(K));
            if (OP == 10) asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : : "v"(w[i]), "v"(w[(i + 1) % kChains]), "v"(a[i]), "v"(b[i]) : "vcc");
            if (OP == 11) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(K));
            if (OP == 12) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(K), "v"(b[i]));
            if (OP == 13) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 14) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(b[i]), "v"(K));
            // round 4: the rest of what the build kernels' text is made of (tools/isa_mix.py prices the mix with these)
            if (OP == 15) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]));
            if (OP == 16) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i]));
            if (OP == 17) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 18) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 19) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
            if (OP == 20) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 21) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(K));
            if (OP == 22) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 23) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(K));
            if (OP == 24) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 25) asm volatile("v_cmp_ne_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b[i]) : "vcc");
            // (a lone v_cndmask_b32 reading vcc measured 22 cycles in this harness -- an artefact of the inline-asm constraints, the
            // pair v_cmp + v_cndmask above costs 4.4 per instruction: the lone form is not reported)
            if (OP == 27) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 28) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(w[i]) : "v"(w[(i + 1) % kChains]));
            // the dense scan's table lookups (scan_dense_lut_kernel): an eight-entry byte table per instruction
            if (OP == 29) asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(K));
            if (OP == 30) asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(seed), "v"(K));
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < kChains; ++i) r ^= a[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
    if (r == 0x12345678u) out[0] = r;
}

// the whole hash of one k-mer as the build computes it: canonical minimum, revhash64, bucket + 32-bit mantis
__device__ __forceinline__ uint64_t revhash64(uint64_t x)
{
    x = ((x >> 32) ^ x) * 0xD6E8FEB86659FD93ULL;
    x = ((x >> 32) ^ x) * 0xD6E8FEB86659FD93ULL;
    return (x >> 32) ^ x;
}
__global__ __launch_bounds__(256) void hash_kernel(uint32_t *out, uint32_t seed)
{
    uint64_t s = ((uint64_t)threadIdx.x << 32) | (blockIdx.x * 77u + seed), acc = 0;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < kChains; ++i) {
            s = (s << 2) | ((it + i) & 3u);
            acc += revhash64(s & 0x3fffffffffffffffULL);
        }
    }
    if (acc == 0x12345678u) out[0] = (uint32_t)acc;
}

// LDS operations on a table of `range` 32-bit entries (a power of two), 16 per thread per trip.  The sixteen indices of a
// thread are drawn once, outside the loop, and only rotated by a wave-uniform amount per trip (one v_add + one v_and per
// access: well below what the LDS takes), so that what is timed is the LDS and not the index arithmetic.
// PATTERN 0: random indices (bank conflicts as they come); 1: lane l goes to entry (l + c) mod range -- no conflicts;
// 2: every lane the same entry.
template <int OP, int PATTERN>
__global__ __launch_bounds__(256) void lds_kernel(uint32_t *out, uint32_t range, uint32_t seed)
{
    __shared__ unsigned long long tab[8192];
    for (uint32_t i = threadIdx.x; i < 8192; i += 256) tab[i] = ~0ull;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + seed, r = 0;
    uint32_t *t32 = reinterpret_cast<uint32_t *>(tab);
    const uint32_t mask = range - 1u;
    uint32_t base[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        x = x * 1664525u + 1013904223u;
        base[i] = PATTERN == 0 ? (x >> 11) : PATTERN == 1 ? (threadIdx.x & 63u) + 97u * i : 5u * i;
    }
    for (int it = 0; it < kIters / 16; ++it) {
        const uint32_t rot = (uint32_t)it * 7u;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t idx = (base[i] + rot) & mask;
            if (OP == 0) r += atomicAdd(&t32[idx], 1u);                                   // ds_add_rtn_u32
            if (OP == 1) atomicAdd(&t32[idx], 1u);                                        // ds_add_u32 (no return)
            if (OP == 2) atomicMin(&tab[idx], ((unsigned long long)x << 32) | idx);       // ds_min_u64
            if (OP == 3) atomicMin(&t32[idx], x ^ rot);                                   // ds_min_u32
            if (OP == 4) r += t32[idx];                                                   // ds_read_b32
            if (OP == 5) t32[idx] = x;                                                    // ds_write_b32
        }
    }
    __syncthreads();
    if (r == 0x12345678u || tab[threadIdx.x] == 1234567ull) out[0] = r;
}

static double time_ms(void (*launch)(int, uint32_t *), int blocks, uint32_t *d)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(blocks, d);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(a));
        launch(blocks, d);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    return best;
}

template <int OP> static void launch_valu(int blocks, uint32_t *d) { hipLaunchKernelGGL(valu_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u); }
static void launch_hash(int blocks, uint32_t *d) { hipLaunchKernelGGL(hash_kernel, dim3(blocks), dim3(256), 0, 0, d, 1u); }
static uint32_t g_range = 128;
template <int OP, int PATTERN> static void launch_lds(int blocks, uint32_t *d) { hipLaunchKernelGGL((lds_kernel<OP, PATTERN>), dim3(blocks), dim3(256), 0, 0, d, g_range, 1u); }

int main()
{
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const double clk = p.clockRate * 1e3;     // Hz
    printf("device %s: %d CUs, clock %.0f MHz\n", p.name, cus, clk / 1e6);
    uint32_t *d;
    CK(hipMalloc(&d, 64));
    const char *names[] = {"v_add_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_lshlrev_b64", "v_alignbit_b32", "v_xor_b32",
                           "v_bfe_u32", "v_ffbh_u32", "v_and_or_b32", "v_cmp_lt_u64+cndmask", "v_mul_u32_u24", "v_mad_u32_u24",
                           "v_lshl_add_u32", "v_bitop3_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_and_b32", "v_or_b32", "v_mov_b32",
                           "v_sub_u32", "v_add3_u32", "v_lshl_or_b32", "v_or3_b32", "v_min_u32", "v_cmp_ne_u32", "v_mbcnt_lo_u32_b32", "v_lshl_add_u64",
                           "v_perm_b32", "v_perm_b32 (one SGPR)"};
    void (*fn[])(int, uint32_t *) = {launch_valu<0>, launch_valu<1>, launch_valu<2>, launch_valu<3>, launch_valu<4>, launch_valu<5>, launch_valu<6>,
                                     launch_valu<7>, launch_valu<8>, launch_valu<9>, launch_valu<10>, launch_valu<11>, launch_valu<12>,
                                     launch_valu<13>, launch_valu<14>, launch_valu<15>, launch_valu<16>, launch_valu<17>, launch_valu<18>,
                                     launch_valu<19>, launch_valu<20>, launch_valu<21>, launch_valu<22>, launch_valu<23>, launch_valu<24>,
                                     launch_valu<25>, launch_valu<27>, launch_valu<28>, launch_valu<29>, launch_valu<30>};
    constexpr int kOps = 30;
    printf("%-24s %10s %10s %10s %10s   (cycles per wave-instruction per SIMD at 1, 2, 4, 8 waves per SIMD)\n", "op", "1", "2", "4", "8");
    for (int op = 0; op < kOps; ++op) {
        printf("%-24s", names[op]);
        for (int wps = 1; wps <= 8; wps *= 2) {
            // one 256-thread block = one wave per SIMD of a CU; wps blocks per CU
            const int blocks = cus * wps;
            const double ms = time_ms(fn[op], blocks, d);
            const double inst_per_simd = (double)kIters * kChains * wps * (op == 10 ? 2 : 1);
            printf(" %10.2f", ms * 1e-3 * clk / inst_per_simd);
        }
        printf("\n");
    }
    for (int wps = 1; wps <= 8; wps *= 2) {
        const double ms = time_ms(launch_hash, cus * wps, d);
        const double kmers = (double)cus * wps * 256 * kIters * kChains;
        printf("revhash64 alone, %d waves/SIMD: %.3e k-mers/s  (%.1f cycles per k-mer-wave per SIMD)\n", wps, kmers / (ms * 1e-3),
               ms * 1e-3 * clk / ((double)kIters * kChains * wps));
    }
    const char *lnames[] = {"ds_add_rtn_u32", "ds_add_u32", "ds_min_u64", "ds_min_u32", "ds_read_b32", "ds_write_b32"};
    void (*lfn[3][6])(int, uint32_t *) = {
        {launch_lds<0, 0>, launch_lds<1, 0>, launch_lds<2, 0>, launch_lds<3, 0>, launch_lds<4, 0>, launch_lds<5, 0>},
        {launch_lds<0, 1>, launch_lds<1, 1>, launch_lds<2, 1>, launch_lds<3, 1>, launch_lds<4, 1>, launch_lds<5, 1>},
        {launch_lds<0, 2>, launch_lds<1, 2>, launch_lds<2, 2>, launch_lds<3, 2>, launch_lds<4, 2>, launch_lds<5, 2>}};
    const char *pnames[] = {"random index", "lane l -> entry l + c (no bank conflicts)", "one entry for all lanes"};
    for (int pat = 0; pat < 3; ++pat)
        for (uint32_t range : {128u, 4096u, 8192u}) {
            if (pat && range != 4096u) continue;
            g_range = range;
            printf("LDS, %s in [0, %u): cycles per wave-instruction per CU at 4, 8, 16, 32 waves per CU\n", pnames[pat], range);
            for (int op = 0; op < 6; ++op) {
                printf("  %-16s", lnames[op]);
                for (int wpc = 4; wpc <= 32; wpc *= 2) {
                    const int blocks = cus * (wpc / 4);
                    const double ms = time_ms(lfn[pat][op], blocks, d);
                    const double inst_per_cu = (double)kIters * wpc;
                    printf(" %10.2f", ms * 1e-3 * clk / inst_per_cu);
                }
                printf("\n");
            }
        }
    CK(hipFree(d));
    return 0;
}
