// The DMA-fed 1x1 GEMM launcher (see conv_gemm_kernel.h).
#include "conv_gemm_kernel.h"

namespace loco {

template <int TM>
static void launch_gemm_tm(const ConvArgs& a, const unsigned char* rec, hipStream_t st) {
    constexpr int MT = 64 * TM, SLOTB = MT * 64 + GM_NT * 64;
    size_t lds = (size_t)GM_NSLOT * SLOTB;
    const size_t stage_bytes = (size_t)64 * GM_NT * 4;          // epilogue staging tile
    if (lds < stage_bytes) lds = stage_bytes;
    auto kern = &conv_gemm_bf16x3<TM>;
    static DeviceOnce once;
    if (first_on_device(once))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int HW = a.Hout * a.Wout;
    dim3 grid((HW / GM_NT) * ((a.Cout + MT - 1) / MT) * a.B * a.nsplit);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, a, rec);
}

void launch_conv_gemm(const ConvArgs& a, hipStream_t st) {
    const int HW = a.Hout * a.Wout;
    const size_t rec_floats = (size_t)a.B * a.Cin * HW;
    unsigned char* rec = reinterpret_cast<unsigned char*>(a.partial + (a.partial_floats - rec_floats));
    dim3 g(HW / 256, a.Cin / BKC, a.B);
    if (a.mode == CM_NONE)
        hipLaunchKernelGGL(act_split_kernel<CM_NONE>, g, dim3(256), 0, st, a.in, a.in_bs, HW, a.sc, a.sh, a.scsh_bs, rec);
    else
        hipLaunchKernelGGL(act_split_kernel<CM_GN>, g, dim3(256), 0, st, a.in, a.in_bs, HW, a.sc, a.sh, a.scsh_bs, rec);
    if (a.gemm_tm == 4) launch_gemm_tm<4>(a, rec, st);
    else launch_gemm_tm<2>(a, rec, st);
}

}  // namespace loco
