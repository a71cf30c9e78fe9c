// upx_reg_fused_single.hip - fused streaming kernels for launches that carry ONE band (no second gain slot: fewer
// registers, no spills; the mask through mask_weight) and their live-slot specialisations (upx::Live, upx_kernels.h).
#include "upx_kernels.h"

namespace upxk {
const KernelEntry* find_kernel_single(int log2n, int k, int variant) {
    static const std::map<std::tuple<int, int, int>, KernelEntry> table = [] {
        std::map<std::tuple<int, int, int>, KernelEntry> t;
        // variant 10 = variant 0 for launches that carry a single band (no second gain slot: fewer registers, no spills)
#define UPX_REG1(L, K) \
    t[std::make_tuple(L, K, 10)] = Entry<upx::Cfg<L, K, 16>, 2, false>::get("upx_band_kernel<upx::Cfg<" #L ", " #K ", 16>, 2, false>");
#define UPX_REG1_WIDE(L, K) \
    t[std::make_tuple(L, K, 10)] = Entry<upx::WideCfg<L, K>, 2, false>::get("upx_band_kernel<upx::WideCfg<" #L ", " #K ">, 2, false>");
        UPX_REG1(8, 2) UPX_REG1(9, 2) UPX_REG1(10, 2) UPX_REG1(11, 2) UPX_REG1(8, 4) UPX_REG1(9, 4) UPX_REG1(10, 4) UPX_REG1(11, 4)
        UPX_REG1(8, 8) UPX_REG1(9, 8) UPX_REG1(10, 8) UPX_REG1(11, 8)
        UPX_REG1_WIDE(12, 2) UPX_REG1_WIDE(12, 4) UPX_REG1_WIDE(12, 8) UPX_REG1_WIDE(13, 2) UPX_REG1_WIDE(13, 4) UPX_REG1_WIDE(13, 8)
        // variant 100 + 10 S0 + S1 = variant 10 specialised for the live own-bin slots [S0, S1) (upx::Live; hop N/4).
        // The reference's planner ties N to the band's low edge, so a middle band covers bins 31..205 whatever its N
        // (slots 0..3 at N = 1024, 0..1 at N = 2048) and a top band everything from bin 31 up (slots 1..7 at N = 256).
        // (Live<1, 8> - a top band, everything but slot 0 - was measured slower than the general flavour at N = 256:
        // 0.395 vs 0.388 ms, one slot in eight pruned against the eager read order of the last forward pass lost)
#define UPX_REGL(L, A, B)                                                                             \
    t[std::make_tuple(L, 4, 100 + 10 * A + B)] = Entry<upx::Cfg<L, 4, 16>, 2, false, upx::Live<A, B>>::get( \
        "upx_band_kernel<upx::Cfg<" #L ", 4, 16>, 2, false, upx::Live<" #A ", " #B ">>");
        UPX_REGL(10, 0, 2) UPX_REGL(10, 0, 3) UPX_REGL(10, 0, 4) UPX_REGL(11, 0, 2) UPX_REGL(11, 0, 3) UPX_REGL(11, 0, 4)
        return t;
    }();
    auto it = table.find(std::make_tuple(log2n, k, variant));
    return it == table.end() ? nullptr : &it->second;
}
}   // namespace upxk
