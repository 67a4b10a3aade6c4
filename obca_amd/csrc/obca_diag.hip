// obca_diag.hip -- libobca_diag.so: a DIAGNOSTIC library of its own (tests, obca_amd.selftest() and bench.py's bit-equality line load it explicitly through
// obca_amd/diag.py; the product library libobca_hip.so does not contain, link or call any of it -- include/obca_diag.h).
//
// obca_diag_leave_pattern: a kernel that leaves a bit pattern in everything a following workgroup inherits from its predecessor on the same SIMD / CU -- mask bit 0: the vector
// registers, bit 1: the accumulation registers, bit 2: the CU's LDS (the 64-bit pattern `value`), bit 3: scratch memory.  A solver must not care: whatever it reads, it has
// written.  Round 5: the parking kernels' results changed when ANOTHER PROCESS shared the GPU -- two stores of the stage assembly had been lost and four LDS words were read
// unwritten.  A NaN pattern found nothing (the words went through fmax()); a large finite one finds it (tests/test_gpu_history.py).
#include <hip/hip_runtime.h>
#include <cstring>
#include "../../include/obca_diag.h"

__global__ __launch_bounds__(64, 1) void obca_dirty_kernel(int mask, unsigned lo, unsigned hi, unsigned *sink) {
    extern __shared__ unsigned dirty_lds[];      // 40 KB: four workgroups cover the 160 KB of a CU
    const unsigned nanw = lo;
    if (mask & 4) for (int i = threadIdx.x; i < 10240; i += 64) dirty_lds[i] = (i & 1) ? hi : lo;
    unsigned acc = 0;
    if (mask & 8) {
        volatile unsigned priv[1024];                 // 4 KB per lane of scratch
        for (int i = 0; i < 1024; i++) priv[(i * 5 + threadIdx.x) & 1023] = nanw;
        acc += priv[threadIdx.x & 1023];
    }
    if (mask & 2) {
        asm volatile("v_accvgpr_write_b32 a0, %0" :: "v"(nanw) : "a0");
        asm volatile("v_accvgpr_write_b32 a1, %0" :: "v"(nanw) : "a1");
        asm volatile("v_accvgpr_write_b32 a2, %0" :: "v"(nanw) : "a2");
        asm volatile("v_accvgpr_write_b32 a3, %0" :: "v"(nanw) : "a3");
        asm volatile("v_accvgpr_write_b32 a4, %0" :: "v"(nanw) : "a4");
        asm volatile("v_accvgpr_write_b32 a5, %0" :: "v"(nanw) : "a5");
        asm volatile("v_accvgpr_write_b32 a6, %0" :: "v"(nanw) : "a6");
        asm volatile("v_accvgpr_write_b32 a7, %0" :: "v"(nanw) : "a7");
        asm volatile("v_accvgpr_write_b32 a8, %0" :: "v"(nanw) : "a8");
        asm volatile("v_accvgpr_write_b32 a9, %0" :: "v"(nanw) : "a9");
        asm volatile("v_accvgpr_write_b32 a10, %0" :: "v"(nanw) : "a10");
        asm volatile("v_accvgpr_write_b32 a11, %0" :: "v"(nanw) : "a11");
        asm volatile("v_accvgpr_write_b32 a12, %0" :: "v"(nanw) : "a12");
        asm volatile("v_accvgpr_write_b32 a13, %0" :: "v"(nanw) : "a13");
        asm volatile("v_accvgpr_write_b32 a14, %0" :: "v"(nanw) : "a14");
        asm volatile("v_accvgpr_write_b32 a15, %0" :: "v"(nanw) : "a15");
        asm volatile("v_accvgpr_write_b32 a16, %0" :: "v"(nanw) : "a16");
        asm volatile("v_accvgpr_write_b32 a17, %0" :: "v"(nanw) : "a17");
        asm volatile("v_accvgpr_write_b32 a18, %0" :: "v"(nanw) : "a18");
        asm volatile("v_accvgpr_write_b32 a19, %0" :: "v"(nanw) : "a19");
        asm volatile("v_accvgpr_write_b32 a20, %0" :: "v"(nanw) : "a20");
        asm volatile("v_accvgpr_write_b32 a21, %0" :: "v"(nanw) : "a21");
        asm volatile("v_accvgpr_write_b32 a22, %0" :: "v"(nanw) : "a22");
        asm volatile("v_accvgpr_write_b32 a23, %0" :: "v"(nanw) : "a23");
        asm volatile("v_accvgpr_write_b32 a24, %0" :: "v"(nanw) : "a24");
        asm volatile("v_accvgpr_write_b32 a25, %0" :: "v"(nanw) : "a25");
        asm volatile("v_accvgpr_write_b32 a26, %0" :: "v"(nanw) : "a26");
        asm volatile("v_accvgpr_write_b32 a27, %0" :: "v"(nanw) : "a27");
        asm volatile("v_accvgpr_write_b32 a28, %0" :: "v"(nanw) : "a28");
        asm volatile("v_accvgpr_write_b32 a29, %0" :: "v"(nanw) : "a29");
        asm volatile("v_accvgpr_write_b32 a30, %0" :: "v"(nanw) : "a30");
        asm volatile("v_accvgpr_write_b32 a31, %0" :: "v"(nanw) : "a31");
        asm volatile("v_accvgpr_write_b32 a32, %0" :: "v"(nanw) : "a32");
        asm volatile("v_accvgpr_write_b32 a33, %0" :: "v"(nanw) : "a33");
        asm volatile("v_accvgpr_write_b32 a34, %0" :: "v"(nanw) : "a34");
        asm volatile("v_accvgpr_write_b32 a35, %0" :: "v"(nanw) : "a35");
        asm volatile("v_accvgpr_write_b32 a36, %0" :: "v"(nanw) : "a36");
        asm volatile("v_accvgpr_write_b32 a37, %0" :: "v"(nanw) : "a37");
        asm volatile("v_accvgpr_write_b32 a38, %0" :: "v"(nanw) : "a38");
        asm volatile("v_accvgpr_write_b32 a39, %0" :: "v"(nanw) : "a39");
        asm volatile("v_accvgpr_write_b32 a40, %0" :: "v"(nanw) : "a40");
        asm volatile("v_accvgpr_write_b32 a41, %0" :: "v"(nanw) : "a41");
        asm volatile("v_accvgpr_write_b32 a42, %0" :: "v"(nanw) : "a42");
        asm volatile("v_accvgpr_write_b32 a43, %0" :: "v"(nanw) : "a43");
        asm volatile("v_accvgpr_write_b32 a44, %0" :: "v"(nanw) : "a44");
        asm volatile("v_accvgpr_write_b32 a45, %0" :: "v"(nanw) : "a45");
        asm volatile("v_accvgpr_write_b32 a46, %0" :: "v"(nanw) : "a46");
        asm volatile("v_accvgpr_write_b32 a47, %0" :: "v"(nanw) : "a47");
        asm volatile("v_accvgpr_write_b32 a48, %0" :: "v"(nanw) : "a48");
        asm volatile("v_accvgpr_write_b32 a49, %0" :: "v"(nanw) : "a49");
        asm volatile("v_accvgpr_write_b32 a50, %0" :: "v"(nanw) : "a50");
        asm volatile("v_accvgpr_write_b32 a51, %0" :: "v"(nanw) : "a51");
        asm volatile("v_accvgpr_write_b32 a52, %0" :: "v"(nanw) : "a52");
        asm volatile("v_accvgpr_write_b32 a53, %0" :: "v"(nanw) : "a53");
        asm volatile("v_accvgpr_write_b32 a54, %0" :: "v"(nanw) : "a54");
        asm volatile("v_accvgpr_write_b32 a55, %0" :: "v"(nanw) : "a55");
        asm volatile("v_accvgpr_write_b32 a56, %0" :: "v"(nanw) : "a56");
        asm volatile("v_accvgpr_write_b32 a57, %0" :: "v"(nanw) : "a57");
        asm volatile("v_accvgpr_write_b32 a58, %0" :: "v"(nanw) : "a58");
        asm volatile("v_accvgpr_write_b32 a59, %0" :: "v"(nanw) : "a59");
        asm volatile("v_accvgpr_write_b32 a60, %0" :: "v"(nanw) : "a60");
        asm volatile("v_accvgpr_write_b32 a61, %0" :: "v"(nanw) : "a61");
        asm volatile("v_accvgpr_write_b32 a62, %0" :: "v"(nanw) : "a62");
        asm volatile("v_accvgpr_write_b32 a63, %0" :: "v"(nanw) : "a63");
        asm volatile("v_accvgpr_write_b32 a64, %0" :: "v"(nanw) : "a64");
        asm volatile("v_accvgpr_write_b32 a65, %0" :: "v"(nanw) : "a65");
        asm volatile("v_accvgpr_write_b32 a66, %0" :: "v"(nanw) : "a66");
        asm volatile("v_accvgpr_write_b32 a67, %0" :: "v"(nanw) : "a67");
        asm volatile("v_accvgpr_write_b32 a68, %0" :: "v"(nanw) : "a68");
        asm volatile("v_accvgpr_write_b32 a69, %0" :: "v"(nanw) : "a69");
        asm volatile("v_accvgpr_write_b32 a70, %0" :: "v"(nanw) : "a70");
        asm volatile("v_accvgpr_write_b32 a71, %0" :: "v"(nanw) : "a71");
        asm volatile("v_accvgpr_write_b32 a72, %0" :: "v"(nanw) : "a72");
        asm volatile("v_accvgpr_write_b32 a73, %0" :: "v"(nanw) : "a73");
        asm volatile("v_accvgpr_write_b32 a74, %0" :: "v"(nanw) : "a74");
        asm volatile("v_accvgpr_write_b32 a75, %0" :: "v"(nanw) : "a75");
        asm volatile("v_accvgpr_write_b32 a76, %0" :: "v"(nanw) : "a76");
        asm volatile("v_accvgpr_write_b32 a77, %0" :: "v"(nanw) : "a77");
        asm volatile("v_accvgpr_write_b32 a78, %0" :: "v"(nanw) : "a78");
        asm volatile("v_accvgpr_write_b32 a79, %0" :: "v"(nanw) : "a79");
        asm volatile("v_accvgpr_write_b32 a80, %0" :: "v"(nanw) : "a80");
        asm volatile("v_accvgpr_write_b32 a81, %0" :: "v"(nanw) : "a81");
        asm volatile("v_accvgpr_write_b32 a82, %0" :: "v"(nanw) : "a82");
        asm volatile("v_accvgpr_write_b32 a83, %0" :: "v"(nanw) : "a83");
        asm volatile("v_accvgpr_write_b32 a84, %0" :: "v"(nanw) : "a84");
        asm volatile("v_accvgpr_write_b32 a85, %0" :: "v"(nanw) : "a85");
        asm volatile("v_accvgpr_write_b32 a86, %0" :: "v"(nanw) : "a86");
        asm volatile("v_accvgpr_write_b32 a87, %0" :: "v"(nanw) : "a87");
        asm volatile("v_accvgpr_write_b32 a88, %0" :: "v"(nanw) : "a88");
        asm volatile("v_accvgpr_write_b32 a89, %0" :: "v"(nanw) : "a89");
        asm volatile("v_accvgpr_write_b32 a90, %0" :: "v"(nanw) : "a90");
        asm volatile("v_accvgpr_write_b32 a91, %0" :: "v"(nanw) : "a91");
        asm volatile("v_accvgpr_write_b32 a92, %0" :: "v"(nanw) : "a92");
        asm volatile("v_accvgpr_write_b32 a93, %0" :: "v"(nanw) : "a93");
        asm volatile("v_accvgpr_write_b32 a94, %0" :: "v"(nanw) : "a94");
        asm volatile("v_accvgpr_write_b32 a95, %0" :: "v"(nanw) : "a95");
        asm volatile("v_accvgpr_write_b32 a96, %0" :: "v"(nanw) : "a96");
        asm volatile("v_accvgpr_write_b32 a97, %0" :: "v"(nanw) : "a97");
        asm volatile("v_accvgpr_write_b32 a98, %0" :: "v"(nanw) : "a98");
        asm volatile("v_accvgpr_write_b32 a99, %0" :: "v"(nanw) : "a99");
        asm volatile("v_accvgpr_write_b32 a100, %0" :: "v"(nanw) : "a100");
        asm volatile("v_accvgpr_write_b32 a101, %0" :: "v"(nanw) : "a101");
        asm volatile("v_accvgpr_write_b32 a102, %0" :: "v"(nanw) : "a102");
        asm volatile("v_accvgpr_write_b32 a103, %0" :: "v"(nanw) : "a103");
        asm volatile("v_accvgpr_write_b32 a104, %0" :: "v"(nanw) : "a104");
        asm volatile("v_accvgpr_write_b32 a105, %0" :: "v"(nanw) : "a105");
        asm volatile("v_accvgpr_write_b32 a106, %0" :: "v"(nanw) : "a106");
        asm volatile("v_accvgpr_write_b32 a107, %0" :: "v"(nanw) : "a107");
        asm volatile("v_accvgpr_write_b32 a108, %0" :: "v"(nanw) : "a108");
        asm volatile("v_accvgpr_write_b32 a109, %0" :: "v"(nanw) : "a109");
        asm volatile("v_accvgpr_write_b32 a110, %0" :: "v"(nanw) : "a110");
        asm volatile("v_accvgpr_write_b32 a111, %0" :: "v"(nanw) : "a111");
        asm volatile("v_accvgpr_write_b32 a112, %0" :: "v"(nanw) : "a112");
        asm volatile("v_accvgpr_write_b32 a113, %0" :: "v"(nanw) : "a113");
        asm volatile("v_accvgpr_write_b32 a114, %0" :: "v"(nanw) : "a114");
        asm volatile("v_accvgpr_write_b32 a115, %0" :: "v"(nanw) : "a115");
        asm volatile("v_accvgpr_write_b32 a116, %0" :: "v"(nanw) : "a116");
        asm volatile("v_accvgpr_write_b32 a117, %0" :: "v"(nanw) : "a117");
        asm volatile("v_accvgpr_write_b32 a118, %0" :: "v"(nanw) : "a118");
        asm volatile("v_accvgpr_write_b32 a119, %0" :: "v"(nanw) : "a119");
        asm volatile("v_accvgpr_write_b32 a120, %0" :: "v"(nanw) : "a120");
        asm volatile("v_accvgpr_write_b32 a121, %0" :: "v"(nanw) : "a121");
        asm volatile("v_accvgpr_write_b32 a122, %0" :: "v"(nanw) : "a122");
        asm volatile("v_accvgpr_write_b32 a123, %0" :: "v"(nanw) : "a123");
        asm volatile("v_accvgpr_write_b32 a124, %0" :: "v"(nanw) : "a124");
        asm volatile("v_accvgpr_write_b32 a125, %0" :: "v"(nanw) : "a125");
        asm volatile("v_accvgpr_write_b32 a126, %0" :: "v"(nanw) : "a126");
        asm volatile("v_accvgpr_write_b32 a127, %0" :: "v"(nanw) : "a127");
        asm volatile("v_accvgpr_write_b32 a128, %0" :: "v"(nanw) : "a128");
        asm volatile("v_accvgpr_write_b32 a129, %0" :: "v"(nanw) : "a129");
        asm volatile("v_accvgpr_write_b32 a130, %0" :: "v"(nanw) : "a130");
        asm volatile("v_accvgpr_write_b32 a131, %0" :: "v"(nanw) : "a131");
        asm volatile("v_accvgpr_write_b32 a132, %0" :: "v"(nanw) : "a132");
        asm volatile("v_accvgpr_write_b32 a133, %0" :: "v"(nanw) : "a133");
        asm volatile("v_accvgpr_write_b32 a134, %0" :: "v"(nanw) : "a134");
        asm volatile("v_accvgpr_write_b32 a135, %0" :: "v"(nanw) : "a135");
        asm volatile("v_accvgpr_write_b32 a136, %0" :: "v"(nanw) : "a136");
        asm volatile("v_accvgpr_write_b32 a137, %0" :: "v"(nanw) : "a137");
        asm volatile("v_accvgpr_write_b32 a138, %0" :: "v"(nanw) : "a138");
        asm volatile("v_accvgpr_write_b32 a139, %0" :: "v"(nanw) : "a139");
        asm volatile("v_accvgpr_write_b32 a140, %0" :: "v"(nanw) : "a140");
        asm volatile("v_accvgpr_write_b32 a141, %0" :: "v"(nanw) : "a141");
        asm volatile("v_accvgpr_write_b32 a142, %0" :: "v"(nanw) : "a142");
        asm volatile("v_accvgpr_write_b32 a143, %0" :: "v"(nanw) : "a143");
        asm volatile("v_accvgpr_write_b32 a144, %0" :: "v"(nanw) : "a144");
        asm volatile("v_accvgpr_write_b32 a145, %0" :: "v"(nanw) : "a145");
        asm volatile("v_accvgpr_write_b32 a146, %0" :: "v"(nanw) : "a146");
        asm volatile("v_accvgpr_write_b32 a147, %0" :: "v"(nanw) : "a147");
        asm volatile("v_accvgpr_write_b32 a148, %0" :: "v"(nanw) : "a148");
        asm volatile("v_accvgpr_write_b32 a149, %0" :: "v"(nanw) : "a149");
        asm volatile("v_accvgpr_write_b32 a150, %0" :: "v"(nanw) : "a150");
        asm volatile("v_accvgpr_write_b32 a151, %0" :: "v"(nanw) : "a151");
        asm volatile("v_accvgpr_write_b32 a152, %0" :: "v"(nanw) : "a152");
        asm volatile("v_accvgpr_write_b32 a153, %0" :: "v"(nanw) : "a153");
        asm volatile("v_accvgpr_write_b32 a154, %0" :: "v"(nanw) : "a154");
        asm volatile("v_accvgpr_write_b32 a155, %0" :: "v"(nanw) : "a155");
        asm volatile("v_accvgpr_write_b32 a156, %0" :: "v"(nanw) : "a156");
        asm volatile("v_accvgpr_write_b32 a157, %0" :: "v"(nanw) : "a157");
        asm volatile("v_accvgpr_write_b32 a158, %0" :: "v"(nanw) : "a158");
        asm volatile("v_accvgpr_write_b32 a159, %0" :: "v"(nanw) : "a159");
        asm volatile("v_accvgpr_write_b32 a160, %0" :: "v"(nanw) : "a160");
        asm volatile("v_accvgpr_write_b32 a161, %0" :: "v"(nanw) : "a161");
        asm volatile("v_accvgpr_write_b32 a162, %0" :: "v"(nanw) : "a162");
        asm volatile("v_accvgpr_write_b32 a163, %0" :: "v"(nanw) : "a163");
        asm volatile("v_accvgpr_write_b32 a164, %0" :: "v"(nanw) : "a164");
        asm volatile("v_accvgpr_write_b32 a165, %0" :: "v"(nanw) : "a165");
        asm volatile("v_accvgpr_write_b32 a166, %0" :: "v"(nanw) : "a166");
        asm volatile("v_accvgpr_write_b32 a167, %0" :: "v"(nanw) : "a167");
        asm volatile("v_accvgpr_write_b32 a168, %0" :: "v"(nanw) : "a168");
        asm volatile("v_accvgpr_write_b32 a169, %0" :: "v"(nanw) : "a169");
        asm volatile("v_accvgpr_write_b32 a170, %0" :: "v"(nanw) : "a170");
        asm volatile("v_accvgpr_write_b32 a171, %0" :: "v"(nanw) : "a171");
        asm volatile("v_accvgpr_write_b32 a172, %0" :: "v"(nanw) : "a172");
        asm volatile("v_accvgpr_write_b32 a173, %0" :: "v"(nanw) : "a173");
        asm volatile("v_accvgpr_write_b32 a174, %0" :: "v"(nanw) : "a174");
        asm volatile("v_accvgpr_write_b32 a175, %0" :: "v"(nanw) : "a175");
        asm volatile("v_accvgpr_write_b32 a176, %0" :: "v"(nanw) : "a176");
        asm volatile("v_accvgpr_write_b32 a177, %0" :: "v"(nanw) : "a177");
        asm volatile("v_accvgpr_write_b32 a178, %0" :: "v"(nanw) : "a178");
        asm volatile("v_accvgpr_write_b32 a179, %0" :: "v"(nanw) : "a179");
        asm volatile("v_accvgpr_write_b32 a180, %0" :: "v"(nanw) : "a180");
        asm volatile("v_accvgpr_write_b32 a181, %0" :: "v"(nanw) : "a181");
        asm volatile("v_accvgpr_write_b32 a182, %0" :: "v"(nanw) : "a182");
        asm volatile("v_accvgpr_write_b32 a183, %0" :: "v"(nanw) : "a183");
        asm volatile("v_accvgpr_write_b32 a184, %0" :: "v"(nanw) : "a184");
        asm volatile("v_accvgpr_write_b32 a185, %0" :: "v"(nanw) : "a185");
        asm volatile("v_accvgpr_write_b32 a186, %0" :: "v"(nanw) : "a186");
        asm volatile("v_accvgpr_write_b32 a187, %0" :: "v"(nanw) : "a187");
        asm volatile("v_accvgpr_write_b32 a188, %0" :: "v"(nanw) : "a188");
        asm volatile("v_accvgpr_write_b32 a189, %0" :: "v"(nanw) : "a189");
        asm volatile("v_accvgpr_write_b32 a190, %0" :: "v"(nanw) : "a190");
        asm volatile("v_accvgpr_write_b32 a191, %0" :: "v"(nanw) : "a191");
        asm volatile("v_accvgpr_write_b32 a192, %0" :: "v"(nanw) : "a192");
        asm volatile("v_accvgpr_write_b32 a193, %0" :: "v"(nanw) : "a193");
        asm volatile("v_accvgpr_write_b32 a194, %0" :: "v"(nanw) : "a194");
        asm volatile("v_accvgpr_write_b32 a195, %0" :: "v"(nanw) : "a195");
        asm volatile("v_accvgpr_write_b32 a196, %0" :: "v"(nanw) : "a196");
        asm volatile("v_accvgpr_write_b32 a197, %0" :: "v"(nanw) : "a197");
        asm volatile("v_accvgpr_write_b32 a198, %0" :: "v"(nanw) : "a198");
        asm volatile("v_accvgpr_write_b32 a199, %0" :: "v"(nanw) : "a199");
        asm volatile("v_accvgpr_write_b32 a200, %0" :: "v"(nanw) : "a200");
        asm volatile("v_accvgpr_write_b32 a201, %0" :: "v"(nanw) : "a201");
        asm volatile("v_accvgpr_write_b32 a202, %0" :: "v"(nanw) : "a202");
        asm volatile("v_accvgpr_write_b32 a203, %0" :: "v"(nanw) : "a203");
        asm volatile("v_accvgpr_write_b32 a204, %0" :: "v"(nanw) : "a204");
        asm volatile("v_accvgpr_write_b32 a205, %0" :: "v"(nanw) : "a205");
        asm volatile("v_accvgpr_write_b32 a206, %0" :: "v"(nanw) : "a206");
        asm volatile("v_accvgpr_write_b32 a207, %0" :: "v"(nanw) : "a207");
        asm volatile("v_accvgpr_write_b32 a208, %0" :: "v"(nanw) : "a208");
        asm volatile("v_accvgpr_write_b32 a209, %0" :: "v"(nanw) : "a209");
        asm volatile("v_accvgpr_write_b32 a210, %0" :: "v"(nanw) : "a210");
        asm volatile("v_accvgpr_write_b32 a211, %0" :: "v"(nanw) : "a211");
        asm volatile("v_accvgpr_write_b32 a212, %0" :: "v"(nanw) : "a212");
        asm volatile("v_accvgpr_write_b32 a213, %0" :: "v"(nanw) : "a213");
        asm volatile("v_accvgpr_write_b32 a214, %0" :: "v"(nanw) : "a214");
        asm volatile("v_accvgpr_write_b32 a215, %0" :: "v"(nanw) : "a215");
        asm volatile("v_accvgpr_write_b32 a216, %0" :: "v"(nanw) : "a216");
        asm volatile("v_accvgpr_write_b32 a217, %0" :: "v"(nanw) : "a217");
        asm volatile("v_accvgpr_write_b32 a218, %0" :: "v"(nanw) : "a218");
        asm volatile("v_accvgpr_write_b32 a219, %0" :: "v"(nanw) : "a219");
        asm volatile("v_accvgpr_write_b32 a220, %0" :: "v"(nanw) : "a220");
        asm volatile("v_accvgpr_write_b32 a221, %0" :: "v"(nanw) : "a221");
        asm volatile("v_accvgpr_write_b32 a222, %0" :: "v"(nanw) : "a222");
        asm volatile("v_accvgpr_write_b32 a223, %0" :: "v"(nanw) : "a223");
        asm volatile("v_accvgpr_write_b32 a224, %0" :: "v"(nanw) : "a224");
        asm volatile("v_accvgpr_write_b32 a225, %0" :: "v"(nanw) : "a225");
        asm volatile("v_accvgpr_write_b32 a226, %0" :: "v"(nanw) : "a226");
        asm volatile("v_accvgpr_write_b32 a227, %0" :: "v"(nanw) : "a227");
        asm volatile("v_accvgpr_write_b32 a228, %0" :: "v"(nanw) : "a228");
        asm volatile("v_accvgpr_write_b32 a229, %0" :: "v"(nanw) : "a229");
        asm volatile("v_accvgpr_write_b32 a230, %0" :: "v"(nanw) : "a230");
        asm volatile("v_accvgpr_write_b32 a231, %0" :: "v"(nanw) : "a231");
        asm volatile("v_accvgpr_write_b32 a232, %0" :: "v"(nanw) : "a232");
        asm volatile("v_accvgpr_write_b32 a233, %0" :: "v"(nanw) : "a233");
        asm volatile("v_accvgpr_write_b32 a234, %0" :: "v"(nanw) : "a234");
        asm volatile("v_accvgpr_write_b32 a235, %0" :: "v"(nanw) : "a235");
        asm volatile("v_accvgpr_write_b32 a236, %0" :: "v"(nanw) : "a236");
        asm volatile("v_accvgpr_write_b32 a237, %0" :: "v"(nanw) : "a237");
        asm volatile("v_accvgpr_write_b32 a238, %0" :: "v"(nanw) : "a238");
        asm volatile("v_accvgpr_write_b32 a239, %0" :: "v"(nanw) : "a239");
        asm volatile("v_accvgpr_write_b32 a240, %0" :: "v"(nanw) : "a240");
        asm volatile("v_accvgpr_write_b32 a241, %0" :: "v"(nanw) : "a241");
        asm volatile("v_accvgpr_write_b32 a242, %0" :: "v"(nanw) : "a242");
        asm volatile("v_accvgpr_write_b32 a243, %0" :: "v"(nanw) : "a243");
        asm volatile("v_accvgpr_write_b32 a244, %0" :: "v"(nanw) : "a244");
        asm volatile("v_accvgpr_write_b32 a245, %0" :: "v"(nanw) : "a245");
        asm volatile("v_accvgpr_write_b32 a246, %0" :: "v"(nanw) : "a246");
        asm volatile("v_accvgpr_write_b32 a247, %0" :: "v"(nanw) : "a247");
        asm volatile("v_accvgpr_write_b32 a248, %0" :: "v"(nanw) : "a248");
        asm volatile("v_accvgpr_write_b32 a249, %0" :: "v"(nanw) : "a249");
        asm volatile("v_accvgpr_write_b32 a250, %0" :: "v"(nanw) : "a250");
        asm volatile("v_accvgpr_write_b32 a251, %0" :: "v"(nanw) : "a251");
        asm volatile("v_accvgpr_write_b32 a252, %0" :: "v"(nanw) : "a252");
        asm volatile("v_accvgpr_write_b32 a253, %0" :: "v"(nanw) : "a253");
        asm volatile("v_accvgpr_write_b32 a254, %0" :: "v"(nanw) : "a254");
        asm volatile("v_accvgpr_write_b32 a255, %0" :: "v"(nanw) : "a255");
    }
    if (mask & 1) {
        unsigned v[224];
#pragma unroll
        for (int i = 0; i < 224; i++) { v[i] = nanw; asm volatile("" : "+v"(v[i])); }
#pragma unroll
        for (int i = 0; i < 224; i++) asm volatile("" :: "v"(v[i]));
    }
    if (acc == 12345u) sink[0] = acc;
    // where this workgroup ran: XCC_ID (bits 0..3 of the register) in front of HW_ID's CU / shader-array / shader-engine fields (bits 8..15)
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        sink[64 + blockIdx.x] = ((xcc & 0xf) << 8) | ((hw >> 8) & 0xff);
    }
}


// The pattern reaches a CU's whole LDS only if four of the 40 KB workgroups land on it; the dispatcher usually places them so, nothing guarantees it.  The kernel therefore
// records where its workgroups ran (HW_ID: CU, shader array, shader engine; XCC_ID) and the call reports how many (XCC, SE, SA, CU) units saw how many workgroups.
extern "C" int obca_diag_leave_pattern(int device, int mask, double value, int *units_covered, int *units_with_four) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return -1;
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) return -2;
    const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256, nwg = 4 * cus;
    unsigned *sink = nullptr;
    if (hipMalloc((void **)&sink, (size_t)(64 + nwg) * sizeof(unsigned)) != hipSuccess) return -2;
    if (hipMemset(sink, 0, (size_t)(64 + nwg) * sizeof(unsigned)) != hipSuccess) { (void)hipFree(sink); return -2; }
    if (hipFuncSetAttribute((const void *)obca_dirty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 40960) != hipSuccess) { (void)hipFree(sink); return -2; }
    unsigned long long bits; memcpy(&bits, &value, 8);
    if (value != value) bits = ~0ULL;
    hipLaunchKernelGGL(obca_dirty_kernel, dim3(nwg), dim3(64), 40960, 0, mask, (unsigned)bits, (unsigned)(bits >> 32), sink);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    unsigned *where = new unsigned[nwg];
    if (e == hipSuccess) e = hipMemcpy(where, sink + 64, (size_t)nwg * sizeof(unsigned), hipMemcpyDeviceToHost);
    (void)hipFree(sink);
    int covered = 0, four = 0;
    if (e == hipSuccess) {
        // unit key (12 bits): XCC_ID | SE_ID, SH_ID, CU_ID of HW_ID
        int count[1 << 12]; memset(count, 0, sizeof count);
        for (int i = 0; i < nwg; i++) count[where[i] & 0xfff]++;
        for (int k = 0; k < (1 << 12); k++) { covered += count[k] > 0; four += count[k] >= 4; }
    }
    delete[] where;
    if (units_covered) *units_covered = covered;
    if (units_with_four) *units_with_four = four;
    return e == hipSuccess ? 0 : -2;
}
