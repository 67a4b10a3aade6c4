/*
 * obca_diag.h -- C ABI of libobca_diag.so: DIAGNOSTICS, a library of its own.  Nothing here is part of the drop-in (include/obca_hip.h), nothing in libobca_hip.so
 * refers to it; tests/test_gpu_history.py, obca_amd.selftest() and the bit-equality line of bench.py load it explicitly (obca_amd/diag.py).  No counterpart in the reference.
 */
#ifndef OBCA_DIAG_H
#define OBCA_DIAG_H
#ifdef __cplusplus
extern "C" {
#endif

/* Run a kernel on `device` that fills what a later workgroup inherits from its predecessor on a SIMD / CU with a pattern -- mask bit 0: vector registers, 1: accumulation
 * registers, 2: the CUs' LDS (the 64-bit pattern `value`; NaN = all ones), 3: scratch memory -- and wait for it.  Results of the solves that follow must not depend on it.
 * units_covered / units_with_four (may be NULL): how many compute units ran at least one / at least four of the kernel's 40 KB workgroups (four cover a CU's 160 KB of LDS;
 * the dispatcher usually places them so but nothing guarantees it -- the caller can tell how much of the machine the pattern reached).
 * Returns 0; -1 bad device; -2 device error. */
int obca_diag_leave_pattern(int device, int mask, double value, int *units_covered, int *units_with_four);

#ifdef __cplusplus
}
#endif
#endif
