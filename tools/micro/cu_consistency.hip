// cu_consistency.hip -- does every compute unit of this GPU compute the same bits, run after run?
//
// Round 5 met ONE leased MI355X (of seven) on which the solver's results differed between two runs of the same inputs whenever a batch filled the machine (>= 640
// one-wavefront workgroups), while five other boxes were bit-reproducible over thousands of solves.  (What that box had met: foreign leftovers in LDS, read by two never-stored
// sums of the parking kernels -- DESIGN.md section 11.  The probe stays as a unit check.)  This probe separates a defect of a hardware unit from a defect of the
// code pattern: 1 024 workgroups of one wavefront each (256 VGPRs + 40 KB of LDS: four per CU, one per SIMD, as the solver's) all run the SAME deterministic work on the
// SAME inputs, in five categories that mirror what the solver does; every workgroup reports a checksum per category and the unit it ran on (XCC, SE, CU, SIMD).  Any
// checksum that deviates from the majority names the category and the unit.
//   0 fp64 arithmetic   : fma chains, division, sqrt, v_rcp / v_rsq + Newton, sin, cos, log, pow, exp
//   1 LDS exchange      : lane l writes, wavefront fence + wave barrier (no s_barrier, the solver's LDS_SYNC), lane l reads what lane l + 17 wrote; ds_read_b128; ds_add_f64
//   2 register exchange : DPP (quad_perm, row mirror, wave_shr), ds_bpermute, v_readlane
//   3 HBM exchange      : lane l stores to the workgroup's own 200 KB region, wavefront fence + wave barrier WITHOUT s_waitcnt vmcnt(0), lane l loads what lane l + 17 stored
//                         (the solver's stage records travel like this between the lanes of an instance; the same with a vmcnt(0) drain in between = category 4)
//   4 HBM exchange, drained
//
//   hipcc --offload-arch=gfx950 -O2 -o cu_consistency cu_consistency.hip && ./cu_consistency [repeats]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <map>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define NCAT 5
#define REGION 25600      // doubles of private HBM per workgroup (200 KB, the size of an instance's state)

__device__ inline unsigned long long mix(unsigned long long h, double v) { h ^= (unsigned long long)__double_as_longlong(v); h *= 0x100000001b3ULL; return h ^ (h >> 29); }
template <int CTRL> __device__ inline double dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(64, 1) void probe(int iters, const double *seed, double *region, unsigned long long *out /* per workgroup: NCAT checksums, hw id, xcc id */) {
    extern __shared__ __attribute__((aligned(16))) double lds[];      // 40 KB: four workgroups per CU
    const int lane = threadIdx.x;
    double *mine = region + (size_t)blockIdx.x * REGION;
    unsigned long long h[NCAT];
    for (int c = 0; c < NCAT; c++) h[c] = 0xcbf29ce484222325ULL;
    double x = seed[lane], y = seed[64 + lane];
    for (int i = lane; i < 5120; i += 64) lds[i] = 0.0;      // (nothing below reads LDS it has not written)
    WSYNC();
    for (int it = 0; it < iters; it++) {
        // ---- 0: arithmetic
        {
            double a = x, b = y;
            for (int q = 0; q < 8; q++) { a = fma(a, 0.75, b * 0.125) + 1e-3 * q; b = fma(b, b, a) / (1.0 + a * a); }
            double s, c; sincos(a, &s, &c);
            const double r = __builtin_amdgcn_rcp(1.5 + b * b), rn = r * (2.0 - (1.5 + b * b) * r);
            const double q2 = sqrt(2.0 + a * a) + log(1.5 + b * b) + pow(1.25 + 0.5 * fabs(s), 0.7 + 0.2 * c) + exp(-fabs(a)) + rn;
            h[0] = mix(mix(mix(h[0], a), b), q2);
            x = 0.5 * x + 0.25 * s + 0.01; y = 0.5 * y + 0.25 * c - 0.01;
        }
        // ---- 1: LDS exchange (wave-level ordering only)
        {
            const int base = (it * 131) % 4096;
            lds[base + lane] = x + it; lds[base + 64 + lane] = y - it;
            WSYNC();
            const double u = lds[base + (lane + 17) % 64], v = lds[base + 64 + (lane + 45) % 64];
            const double2 w2 = *(const double2 *)&lds[(base & ~1) + 2 * (lane % 32)];
            WSYNC();
            if (lane % 4 == 0) lds[4600 + lane / 4] = 0.0;
            WSYNC();
            __hip_atomic_fetch_add(&lds[4600 + lane / 4], u * (1 + lane % 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            WSYNC();
            h[1] = mix(mix(mix(mix(h[1], u), v), w2.x + w2.y), lds[4600 + lane / 4]);
            WSYNC();
        }
        // ---- 2: register exchange
        {
            double v = x * (lane + 1);
            v += dpp<0xB1>(v); v += dpp<0x4E>(v); v += dpp<0x141>(v); v += dpp<0x140>(v);
            const double s1 = dpp<0x138>(y), bp = __shfl(x, (lane * 7 + 3) % 64, 64), sx = __shfl_xor(v, 32, 64);
            const double rl = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(y), 41), __builtin_amdgcn_readlane(__double2loint(y), 41));
            h[2] = mix(mix(mix(mix(mix(h[2], v), s1), bp), sx), rl);
        }
        // ---- 3 / 4: HBM exchange inside the wavefront
        for (int drained = 0; drained < 2; drained++) {
            const int base = ((it * 977 + drained * 311) % (REGION / 128 - 1)) * 128;
            mine[base + lane] = x * (it + 1) + drained; mine[base + 64 + lane] = y + lane;
            if (drained) __builtin_amdgcn_s_waitcnt(0x0F70);
            WSYNC();
            const double u = mine[base + (lane + 17) % 64], v = mine[base + 64 + (lane + 45) % 64];
            h[3 + drained] = mix(mix(h[3 + drained], u), v);
            WSYNC();
        }
    }
    // fold the lanes' checksums (xor is order-free)
    for (int c = 0; c < NCAT; c++) {
        unsigned long long v = h[c];
        for (int o = 32; o; o >>= 1) v ^= ((unsigned long long)__shfl_xor((int)(v >> 32), o, 64) << 32) | (unsigned)__shfl_xor((int)v, o, 64);
        if (lane == 0) out[(size_t)blockIdx.x * (NCAT + 2) + c] = v;
    }
    if (lane == 0) {
        out[(size_t)blockIdx.x * (NCAT + 2) + NCAT] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        out[(size_t)blockIdx.x * (NCAT + 2) + NCAT + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 6, NB = 1024, iters = 3000;
    const char *cat[NCAT] = {"fp64 arithmetic", "LDS exchange (wave-level ordering)", "register exchange (DPP / bpermute / readlane)", "HBM exchange, no vmcnt drain", "HBM exchange, drained"};
    std::vector<double> seed(128); srand(3); for (auto &v : seed) v = rand() / (double)RAND_MAX - 0.5;
    double *dseed, *region; unsigned long long *dout;
    CHK(hipMalloc(&dseed, 128 * 8)); CHK(hipMalloc(&region, (size_t)NB * REGION * 8)); CHK(hipMalloc(&dout, (size_t)NB * (NCAT + 2) * 8));
    CHK(hipMemcpy(dseed, seed.data(), 128 * 8, hipMemcpyHostToDevice)); CHK(hipMemset(region, 0, (size_t)NB * REGION * 8));
    CHK(hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 40960));
    std::vector<unsigned long long> o((size_t)NB * (NCAT + 2));
    unsigned long long ref[NCAT]; bool have = false; long bad_total = 0;
    std::map<unsigned long long, int> bad_units;
    for (int r = 0; r < reps; r++) {
        hipLaunchKernelGGL(probe, dim3(NB), dim3(64), 40960, 0, iters, dseed, region, dout);
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(o.data(), dout, o.size() * 8, hipMemcpyDeviceToHost));
        if (!have) {      // reference = the majority value of the first run, per category
            for (int c = 0; c < NCAT; c++) { std::map<unsigned long long, int> cnt; for (int b = 0; b < NB; b++) cnt[o[(size_t)b * (NCAT + 2) + c]]++; int best = 0; for (auto &kv : cnt) if (kv.second > best) { best = kv.second; ref[c] = kv.first; } }
            have = true;
        }
        for (int c = 0; c < NCAT; c++) {
            int nb = 0;
            for (int b = 0; b < NB; b++) if (o[(size_t)b * (NCAT + 2) + c] != ref[c]) {
                const unsigned long long hw = o[(size_t)b * (NCAT + 2) + NCAT], xcc = o[(size_t)b * (NCAT + 2) + NCAT + 1] & 0xF;
                if (nb < 6) printf("  run %d, %s: workgroup %d deviates on xcc%llu se%llu sh%llu cu%llu simd%llu\n", r, cat[c], b, xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3);
                bad_units[(xcc << 16) | (hw & 0xFF00)]++; nb++;
            }
            if (nb) printf("run %d, %-46s: %d of %d workgroups deviate\n", r, cat[c], nb, NB);
            bad_total += nb;
        }
    }
    printf("cu_consistency: %d runs x %d workgroups x %d iterations, %ld deviating (workgroup, category) results", reps, NB, iters, bad_total);
    if (bad_total) { printf("; units:"); for (auto &kv : bad_units) printf(" xcc%llu/se%llu/sh%llu/cu%llu x%d", kv.first >> 16, (kv.first >> 13) & 7, (kv.first >> 12) & 1, (kv.first >> 8) & 15, kv.second); }
    printf("\n");
    return bad_total ? 1 : 0;
}
