// upx_reg_fused.hip - fused streaming kernels plans select by default for MERGED launches (several bands, per-bin gain
// list): 16 points per lane, wide streams for N = 4096 / 8192 (upx_kernels.h).  Single-band flavours: upx_reg_fused_single.hip.
#include "upx_kernels.h"

namespace upxk {
const KernelEntry* find_kernel_single(int log2n, int k, int variant);   // upx_reg_fused_single.hip: variants 10, 100+

const KernelEntry* find_kernel_default(int log2n, int k, int variant) {
    if (variant != 0) return find_kernel_single(log2n, k, variant);
    static const std::map<std::tuple<int, int, int>, KernelEntry> table = [] {
        std::map<std::tuple<int, int, int>, KernelEntry> t;
#define UPX_REG(L, K, PP, W, V) \
    t[std::make_tuple(L, K, V)] = Entry<upx::Cfg<L, K, PP>, W>::get("upx_band_kernel<upx::Cfg<" #L ", " #K ", " #PP ">, " #W ">");
#define UPX_REG_SMALL(K, PP, W, V) \
    UPX_REG(8, K, PP, W, V) UPX_REG(9, K, PP, W, V) UPX_REG(10, K, PP, W, V) UPX_REG(11, K, PP, W, V)
#define UPX_REG_WIDE(L, K) \
    t[std::make_tuple(L, K, 0)] = Entry<upx::WideCfg<L, K>, 2>::get("upx_band_kernel<upx::WideCfg<" #L ", " #K ">, 2>");
        UPX_REG_SMALL(2, 16, 2, 0) UPX_REG_SMALL(4, 16, 2, 0) UPX_REG_SMALL(8, 16, 2, 0)
        UPX_REG_WIDE(12, 2) UPX_REG_WIDE(12, 4) UPX_REG_WIDE(12, 8) UPX_REG_WIDE(13, 2) UPX_REG_WIDE(13, 4) UPX_REG_WIDE(13, 8)
        return t;
    }();
    auto it = table.find(std::make_tuple(log2n, k, variant));
    return it == table.end() ? nullptr : &it->second;
}
}   // namespace upxk
