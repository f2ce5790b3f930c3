// issue_rate.hip - what one vector instruction costs on a gfx950 SIMD, by class and by waves per SIMD.
//
//   hipcc -O3 --offload-arch=gfx950 tools/issue_rate.hip -o gpurun_out/issue_rate
//   gpurun_out/issue_rate profiles/r3_issue_rate.json
//
// the search kernel (k_scale_features) is bound by vector-ALU issue; its roofline in bench.py needs the price
// of an instruction of each class of its mix - measured here, not assumed.  every probe is a loop of 64
// INDEPENDENT instructions of one class (eight rotating destination registers, sources that are never
// written), TRIPS trips; a launch puts `w` such waves on every SIMD of the chip the way the search kernel does -
// one-wave workgroups, 4 w of them per CU, each asking for 160 KB / (4 w) of LDS so that no more fit - and
// time of the launch x maximum clock / (w x instructions per wave)  is the cost of one wave64
// instruction in SIMD cycles.  (the clock held under load can be below the maximum: the figures are upper
// bounds; what matters is the ratio between classes and between columns.)
// mixed probes interleave two classes one to one: do their costs add (one issue port) or overlap?

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int TRIPS = 2048;
constexpr int PER_TRIP = 64;

// eight independent instructions; X(d) expands to one instruction writing register set d
#define EIGHT(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define SIXTYFOUR(X) EIGHT(X) EIGHT(X) EIGHT(X) EIGHT(X) EIGHT(X) EIGHT(X) EIGHT(X) EIGHT(X)


#define PROBE_KERNEL(NAME, BODY)                                                                         \
    __global__ void NAME(uint32_t* out, uint32_t seed)                                                  \
    {                                                                                                   \
        extern __shared__ unsigned char lds[];                                                          \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3u, a2 = a0 ^ 0x55u, a3 = a0 + 7u, a4 = a0 * 5u,    \
                 a5 = a0 | 3u, a6 = a0 + 11u, a7 = a0 * 9u;                                              \
        uint32_t s0 = seed | 1u, s1 = seed + 13u, s2 = 7u;                                               \
        double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;                   \
        double e0 = 1.000001, e1 = 0.999999;                                                            \
        float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7;                    \
        float g0 = 1.0001f, g1 = 0.9999f;                                                               \
        uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3, q4 = a4, q5 = a5, q6 = a6, q7 = a7;                 \
        uint32_t la = (threadIdx.x & 63) * 8, la4 = (threadIdx.x & 63) * 4;                             \
        const unsigned long long lanes = 0x5555555555555555ull + seed;                                  \
        ((uint64_t*)lds)[threadIdx.x & 63] = a0;                                                             \
        __syncthreads();                                                                                \
        _Pragma("nounroll") for (int t = 0; t < TRIPS; ++t) {                                           \
            BODY                                                                                        \
        }                                                                                               \
        uint32_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                              \
        double dr = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;                                               \
        float fr = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;                                                \
        uint64_t qr = q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7;                                             \
        if (r == 0x12345678u && dr == 1.5 && fr == 2.5f && qr == 77u)                                    \
            out[0] = r + s0 + s1 + s2 + (uint32_t)e0 + (uint32_t)e1 + (uint32_t)g0 + (uint32_t)g1 + la; \
    }


// ---- single-class probes ------------------------------------------------------------------------------------
#define I_ADD_U32(N) asm volatile("v_add_u32 %0, %1, %2" : "=v"(a##N) : "v"(s0), "v"(s1));
#define I_ALIGNBIT(N) asm volatile("v_alignbit_b32 %0, %1, %2, 7" : "=v"(a##N) : "v"(s0), "v"(s1));
#define I_ALIGNBIT_V(N) asm volatile("v_alignbit_b32 %0, %1, %2, %3" : "=v"(a##N) : "v"(s0), "v"(s1), "v"(s2));
#define I_BITOP3(N) asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x1e" : "=v"(a##N) : "v"(s0), "v"(s1), "v"(s2));
#define I_MAD_U24(N) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(a##N) : "v"(s0), "v"(s1), "v"(s2));
#define I_MUL_LO(N) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a##N) : "v"(s0), "v"(s1));
#define I_AND(N) asm volatile("v_and_b32 %0, %1, %2" : "=v"(a##N) : "v"(s0), "v"(s1));
#define I_LSHL(N) asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(a##N) : "v"(s0));
#define I_LSHL_ADD(N) asm volatile("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(a##N) : "v"(s0), "v"(s1));
#define I_ADD3(N) asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(a##N) : "v"(s0), "v"(s1), "v"(s2));
#define I_MOV(N) asm volatile("v_mov_b32 %0, %1" : "=v"(a##N) : "v"(s0));
#define I_CNDMASK(N) asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(a##N) : "v"(s0), "v"(s1), "s"(lanes));
#define I_BFE(N) asm volatile("v_bfe_u32 %0, %1, 8, 12" : "=v"(a##N) : "v"(s0));
#define I_OR(N) asm volatile("v_or_b32 %0, %1, %2" : "=v"(a##N) : "v"(s0), "v"(s1));
#define I_SUB_U32(N) asm volatile("v_sub_u32 %0, %1, %2" : "=v"(a##N) : "v"(s0), "v"(s1));
#define I_MIN_U32(N) asm volatile("v_min_u32 %0, %1, %2" : "=v"(a##N) : "v"(s0), "v"(s1));
#define I_MAX_I32(N) asm volatile("v_max_i32 %0, %1, %2" : "=v"(a##N) : "v"(s0), "v"(s1));
#define I_AND_OR(N) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(a##N) : "v"(s0), "v"(s1), "v"(s2));
#define I_LSHL_OR(N) asm volatile("v_lshl_or_b32 %0, %1, 3, %2" : "=v"(a##N) : "v"(s0), "v"(s1));
#define I_PERM(N) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a##N) : "v"(s0), "v"(s1), "v"(s2));
#define I_BCNT(N) asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(a##N) : "v"(s0), "v"(s1));
#define I_BFREV(N) asm volatile("v_bfrev_b32 %0, %1" : "=v"(a##N) : "v"(s0));
#define I_ADDC(N) asm volatile("v_addc_co_u32 %0, vcc, %1, %2, vcc" : "=v"(a##N) : "v"(s0), "v"(s1) : "vcc");
#define I_MUL_U24(N) asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(a##N) : "v"(s0), "v"(s1));
#define I_CVT_I32_F64(N) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(a##N) : "v"(e0));
#define I_FLOOR_F64(N) asm volatile("v_floor_f64 %0, %1" : "=v"(d##N) : "v"(e0));
#define I_MIN_F64(N) asm volatile("v_min_f64 %0, %1, %2" : "=v"(d##N) : "v"(e0), "v"(e1));
#define I_MIN_F32_PLAIN(N) asm volatile("v_min_f32 %0, %1, %2" : "=v"(f##N) : "v"(g0), "v"(g1));
#define I_RSQ_F64(N) asm volatile("v_rsq_f64 %0, %1" : "=v"(d##N) : "v"(e0));
#define I_RCP_F32(N) asm volatile("v_rcp_f32 %0, %1" : "=v"(f##N) : "v"(g0));
#define I_DS_READ_B32_NC(N) asm volatile("ds_read_b32 %0, %1" : "=v"(a##N) : "v"(la4));
#define I_LSHR_B64(N) asm volatile("v_lshrrev_b64 %0, %1, %2" : "=v"(q##N) : "v"(s2), "v"(q##N));
#define I_ADD_F64(N) asm volatile("v_add_f64 %0, %1, %2" : "=v"(d##N) : "v"(e0), "v"(e1));
#define I_MUL_F64(N) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d##N) : "v"(e0), "v"(e1));
#define I_FMA_F64(N) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(d##N) : "v"(e0), "v"(e1), "v"(e0));
#define I_RCP_F64(N) asm volatile("v_rcp_f64 %0, %1" : "=v"(d##N) : "v"(e0));
#define I_CVT_F64_U32(N) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d##N) : "v"(s0));
#define I_CVT_F32_F64(N) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f##N) : "v"(e0));
#define I_CMP_F64(N) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(e0), "v"(e1) : "vcc");
#define I_CMP_F32(N) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(g0), "v"(g1) : "vcc");
#define I_ADD_F32(N) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f##N) : "v"(g0), "v"(g1));
#define I_FMA_F32(N) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f##N) : "v"(g0), "v"(g1), "v"(g0));
#define I_MIN_F32(N) asm volatile("v_min_f32 %0, %1, |%2|" : "=v"(f##N) : "v"(g0), "v"(g1));
#define I_PK_ADD_F32(N) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d##N) : "v"(e0), "v"(e1));
#define I_PK_MUL_F32(N) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d##N) : "v"(e0), "v"(e1));
#define I_PK_FMA_F32(N) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d##N) : "v"(e0), "v"(e1), "v"(e0));
#define I_DS_READ_B64(N) asm volatile("ds_read_b64 %0, %1" : "=v"(q##N) : "v"(la));
#define I_DS_READ_B32(N) asm volatile("ds_read_b32 %0, %1" : "=v"(a##N) : "v"(la));
#define I_READLANE(N) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s2) : "v"(a##N));
#define I_DPP_MIN(N) asm volatile("v_min_i32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xf" : "=v"(a##N) : "v"(s0));
// dependent chain: the latency of a 32-bit op as seen by one wave
#define I_CHAIN_U32(N) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(s1));
#define I_CHAIN_F64(N) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d0) : "v"(e1));
#define WAITLDS asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

PROBE_KERNEL(p_add_u32, SIXTYFOUR(I_ADD_U32))
PROBE_KERNEL(p_alignbit, SIXTYFOUR(I_ALIGNBIT))
PROBE_KERNEL(p_alignbit_v, SIXTYFOUR(I_ALIGNBIT_V))
PROBE_KERNEL(p_bitop3, SIXTYFOUR(I_BITOP3))
PROBE_KERNEL(p_mad_u24, SIXTYFOUR(I_MAD_U24))
PROBE_KERNEL(p_mul_lo, SIXTYFOUR(I_MUL_LO))
PROBE_KERNEL(p_and, SIXTYFOUR(I_AND))
PROBE_KERNEL(p_lshl, SIXTYFOUR(I_LSHL))
PROBE_KERNEL(p_lshl_add, SIXTYFOUR(I_LSHL_ADD))
PROBE_KERNEL(p_add3, SIXTYFOUR(I_ADD3))
PROBE_KERNEL(p_mov, SIXTYFOUR(I_MOV))
PROBE_KERNEL(p_cndmask, SIXTYFOUR(I_CNDMASK))
PROBE_KERNEL(p_bfe, SIXTYFOUR(I_BFE))
PROBE_KERNEL(p_lshr_b64, SIXTYFOUR(I_LSHR_B64))
PROBE_KERNEL(p_or, SIXTYFOUR(I_OR))
PROBE_KERNEL(p_sub_u32, SIXTYFOUR(I_SUB_U32))
PROBE_KERNEL(p_min_u32, SIXTYFOUR(I_MIN_U32))
PROBE_KERNEL(p_max_i32, SIXTYFOUR(I_MAX_I32))
PROBE_KERNEL(p_and_or, SIXTYFOUR(I_AND_OR))
PROBE_KERNEL(p_lshl_or, SIXTYFOUR(I_LSHL_OR))
PROBE_KERNEL(p_perm, SIXTYFOUR(I_PERM))
PROBE_KERNEL(p_bcnt, SIXTYFOUR(I_BCNT))
PROBE_KERNEL(p_bfrev, SIXTYFOUR(I_BFREV))
PROBE_KERNEL(p_addc, SIXTYFOUR(I_ADDC))
PROBE_KERNEL(p_mul_u24, SIXTYFOUR(I_MUL_U24))
PROBE_KERNEL(p_cvt_i32_f64, SIXTYFOUR(I_CVT_I32_F64))
PROBE_KERNEL(p_floor_f64, SIXTYFOUR(I_FLOOR_F64))
PROBE_KERNEL(p_min_f64, SIXTYFOUR(I_MIN_F64))
PROBE_KERNEL(p_min_f32_plain, SIXTYFOUR(I_MIN_F32_PLAIN))
PROBE_KERNEL(p_rsq_f64, SIXTYFOUR(I_RSQ_F64))
PROBE_KERNEL(p_rcp_f32, SIXTYFOUR(I_RCP_F32))
PROBE_KERNEL(p_ds_read_b32_nc, SIXTYFOUR(I_DS_READ_B32_NC) WAITLDS)
PROBE_KERNEL(p_add_f64, SIXTYFOUR(I_ADD_F64))
PROBE_KERNEL(p_mul_f64, SIXTYFOUR(I_MUL_F64))
PROBE_KERNEL(p_fma_f64, SIXTYFOUR(I_FMA_F64))
PROBE_KERNEL(p_rcp_f64, SIXTYFOUR(I_RCP_F64))
PROBE_KERNEL(p_cvt_f64_u32, SIXTYFOUR(I_CVT_F64_U32))
PROBE_KERNEL(p_cvt_f32_f64, SIXTYFOUR(I_CVT_F32_F64))
PROBE_KERNEL(p_cmp_f64, SIXTYFOUR(I_CMP_F64))
PROBE_KERNEL(p_cmp_f32, SIXTYFOUR(I_CMP_F32))
PROBE_KERNEL(p_add_f32, SIXTYFOUR(I_ADD_F32))
PROBE_KERNEL(p_fma_f32, SIXTYFOUR(I_FMA_F32))
PROBE_KERNEL(p_min_f32, SIXTYFOUR(I_MIN_F32))
PROBE_KERNEL(p_pk_add_f32, SIXTYFOUR(I_PK_ADD_F32))
PROBE_KERNEL(p_pk_mul_f32, SIXTYFOUR(I_PK_MUL_F32))
PROBE_KERNEL(p_pk_fma_f32, SIXTYFOUR(I_PK_FMA_F32))
PROBE_KERNEL(p_ds_read_b64, SIXTYFOUR(I_DS_READ_B64) WAITLDS)
PROBE_KERNEL(p_ds_read_b32, SIXTYFOUR(I_DS_READ_B32) WAITLDS)
PROBE_KERNEL(p_readlane, SIXTYFOUR(I_READLANE))
PROBE_KERNEL(p_dpp_min, SIXTYFOUR(I_DPP_MIN))
PROBE_KERNEL(p_chain_u32, SIXTYFOUR(I_CHAIN_U32))
PROBE_KERNEL(p_chain_f64, SIXTYFOUR(I_CHAIN_F64))

// ---- mixed probes: 32 + 32, interleaved one to one ---------------------------------------------------------
#define PAIR(A, B) A(0) B(0) A(1) B(1) A(2) B(2) A(3) B(3) A(4) B(4) A(5) B(5) A(6) B(6) A(7) B(7)
#define MIX64(A, B) PAIR(A, B) PAIR(A, B) PAIR(A, B) PAIR(A, B)
#define I_ADD_U32_B(N) asm volatile("v_add_u32 %0, %1, %2" : "=v"(q##N) : "v"(s0), "v"(s1));
#define I_ALIGNBIT_B(N) asm volatile("v_alignbit_b32 %0, %1, %2, 7" : "=v"(q##N) : "v"(s0), "v"(s1));
#define I_DS_B(N) asm volatile("ds_read_b64 %0, %1" : "=v"(q##N) : "v"(la));
PROBE_KERNEL(p_mix_f64_u32, MIX64(I_ADD_F64, I_ADD_U32))
PROBE_KERNEL(p_mix_f64_alignbit, MIX64(I_ADD_F64, I_ALIGNBIT))
PROBE_KERNEL(p_mix_f64_f32, MIX64(I_ADD_F64, I_ADD_F32))
PROBE_KERNEL(p_mix_u32_f32, MIX64(I_ADD_U32, I_ADD_F32))
PROBE_KERNEL(p_mix_f64_ds, MIX64(I_ADD_F64, I_DS_B) WAITLDS)
PROBE_KERNEL(p_mix_u32_ds, MIX64(I_ADD_U32, I_DS_B) WAITLDS)
// the inclusion test of the search kernel as it stands: add, sub, alignbit on the sign word (dependent inside a
// test, independent between tests)
#define I_TEST_F64(N)                                                                        \
    asm volatile("v_add_f64 %0, %1, %2" : "=v"(d##N) : "v"(e0), "v"(e1));                    \
    asm volatile("v_add_f64 %0, %1, -%0" : "+v"(d##N) : "v"(e0));                            \
    asm volatile("v_alignbit_b32 %0, %0, %1, 25" : "+v"(a##N) : "v"(__double2hiint(d##N)));
PROBE_KERNEL(p_test_f64, EIGHT(I_TEST_F64) EIGHT(I_TEST_F64) EIGHT(I_TEST_F64) EIGHT(I_TEST_F64) EIGHT(I_TEST_F64) EIGHT(I_TEST_F64) EIGHT(I_TEST_F64) EIGHT(I_TEST_F64))

typedef void (*Probe)(uint32_t*, uint32_t);
struct Entry {
    const char* name;
    Probe fn;
    int per_trip;     // wave instructions per trip
};

int main(int argc, char** argv)
{
    const char* json_path = argc > 1 ? argv[1] : nullptr;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    int clock_khz = 0;
    CHECK(hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, 0));
    printf("# %s, %d CUs, max clock %d MHz\n", prop.name, cus, clock_khz / 1000);
    uint32_t* d_out;
    CHECK(hipMalloc(&d_out, 64));
    std::vector<Entry> probes = {
        {"v_add_u32", p_add_u32, 64}, {"v_alignbit_b32 (imm)", p_alignbit, 64}, {"v_alignbit_b32 (vgpr shift)", p_alignbit_v, 64},
        {"v_bitop3_b32", p_bitop3, 64}, {"v_mad_u32_u24", p_mad_u24, 64}, {"v_mul_lo_u32", p_mul_lo, 64},
        {"v_and_b32", p_and, 64}, {"v_lshlrev_b32", p_lshl, 64}, {"v_lshl_add_u32", p_lshl_add, 64}, {"v_add3_u32", p_add3, 64},
        {"v_mov_b32", p_mov, 64}, {"v_cndmask_b32 (sgpr mask)", p_cndmask, 64}, {"v_bfe_u32", p_bfe, 64}, {"v_lshrrev_b64", p_lshr_b64, 64},
        {"v_or_b32", p_or, 64}, {"v_sub_u32", p_sub_u32, 64}, {"v_min_u32", p_min_u32, 64}, {"v_max_i32", p_max_i32, 64},
        {"v_and_or_b32", p_and_or, 64}, {"v_lshl_or_b32", p_lshl_or, 64}, {"v_perm_b32", p_perm, 64}, {"v_bcnt_u32_b32", p_bcnt, 64},
        {"v_bfrev_b32", p_bfrev, 64}, {"v_addc_co_u32", p_addc, 64}, {"v_mul_u32_u24", p_mul_u24, 64},
        {"v_cvt_i32_f64", p_cvt_i32_f64, 64}, {"v_floor_f64", p_floor_f64, 64}, {"v_min_f64", p_min_f64, 64},
        {"v_min_f32", p_min_f32_plain, 64}, {"v_rsq_f64", p_rsq_f64, 64}, {"v_rcp_f32", p_rcp_f32, 64},
        {"ds_read_b32 (4-byte stride: conflict-free)", p_ds_read_b32_nc, 64},
        {"v_add_f64", p_add_f64, 64}, {"v_mul_f64", p_mul_f64, 64}, {"v_fma_f64", p_fma_f64, 64}, {"v_rcp_f64", p_rcp_f64, 64},
        {"v_cvt_f64_u32", p_cvt_f64_u32, 64}, {"v_cvt_f32_f64", p_cvt_f32_f64, 64}, {"v_cmp_lt_f64", p_cmp_f64, 64},
        {"v_cmp_lt_f32", p_cmp_f32, 64}, {"v_add_f32", p_add_f32, 64}, {"v_fma_f32", p_fma_f32, 64}, {"v_min_f32 |abs|", p_min_f32, 64},
        {"v_pk_add_f32", p_pk_add_f32, 64}, {"v_pk_mul_f32", p_pk_mul_f32, 64}, {"v_pk_fma_f32", p_pk_fma_f32, 64},
        {"ds_read_b64", p_ds_read_b64, 64}, {"ds_read_b32 (8-byte stride: 2-way conflict)", p_ds_read_b32, 64}, {"v_readlane_b32", p_readlane, 64},
        {"v_min_i32_dpp", p_dpp_min, 64}, {"chain v_add_u32 (dependent)", p_chain_u32, 64}, {"chain v_add_f64 (dependent)", p_chain_f64, 64},
        {"mix v_add_f64 + v_add_u32", p_mix_f64_u32, 64}, {"mix v_add_f64 + v_alignbit", p_mix_f64_alignbit, 64},
        {"mix v_add_f64 + v_add_f32", p_mix_f64_f32, 64}, {"mix v_add_u32 + v_add_f32", p_mix_u32_f32, 64},
        {"mix v_add_f64 + ds_read_b64", p_mix_f64_ds, 64}, {"mix v_add_u32 + ds_read_b64", p_mix_u32_ds, 64},
        {"test: add_f64, sub_f64, alignbit (x64)", p_test_f64, 192},
    };
    const int waves_per_simd[] = {1, 2, 3, 4, 5, 8};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    FILE* jf = json_path ? fopen(json_path, "w") : nullptr;
    if (jf) fprintf(jf, "{\n \"device\": \"%s\", \"cus\": %d, \"max_clock_mhz\": %d,\n \"unit\": \"SIMD cycles per wave64 instruction at the maximum clock (time of the launch x max clock / (waves per SIMD x instructions per wave)); the clock held under load can be lower, so these are upper bounds - compare classes within a column\",\n \"waves_per_simd\": [1, 2, 3, 4, 5, 8],\n \"cycles\": {\n", prop.name, cus, clock_khz / 1000);
    printf("%-44s", "class \\ waves per SIMD");
    for (int w : waves_per_simd) printf(" %7d", w);
    printf("   (SIMD cycles per wave-instruction at max clock)\n");
    bool first = true;
    for (const Entry& p : probes) {
        printf("%-44s", p.name);
        if (jf) fprintf(jf, "%s  \"%s\": [", first ? "" : ",\n", p.name);
        first = false;
        bool firstw = true;
        for (int w : waves_per_simd) {
            // w waves on every SIMD: one-wave workgroups, 4 w per CU, LDS sized so that no more fit
            const int blocks_per_cu = 4 * w;
            const int threads = 64;
            const size_t lds = (size_t)(160 * 1024 / blocks_per_cu) / 1024 * 1024;
            CHECK(hipFuncSetAttribute((const void*)p.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(p.fn, dim3(cus * blocks_per_cu), dim3(threads), lds, 0, d_out, 12345u + rep);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            const double instr = (double)TRIPS * p.per_trip * w;      // per SIMD
            const double cycles = best * 1e-3 * (clock_khz * 1e3) / instr;
            printf(" %7.2f", cycles);
            if (jf) fprintf(jf, "%s%.3f", firstw ? "" : ", ", cycles);
            firstw = false;
        }
        printf("\n");
        if (jf) fprintf(jf, "]");
    }
    if (jf) {
        fprintf(jf, "\n }\n}\n");
        fclose(jf);
    }
    return 0;
}
