// experiments/upx_exp_fused_p8.hip - fused streaming kernels with 8 points per lane, 4 waves per SIMD (UPX_KERNEL_VARIANT; upx_kernels.h).
#if !defined(UPX_EXPERIMENTS)
#error "experiment kernels: build with -DUPX_EXPERIMENTS (__graft_entry__.build_hip(extra_flags=[\"-DUPX_EXPERIMENTS\"], lib=...)); not part of libupmix_hip.so"
#endif
#include "../upx_kernels.h"

namespace upxk {
const KernelEntry* find_kernel_p8(int log2n, int k) {
    static const std::map<std::tuple<int, int>, KernelEntry> table = [] {
        std::map<std::tuple<int, int>, KernelEntry> t;
#define UPX_REG(L, K) \
    t[std::make_tuple(L, K)] = Entry<upx::Cfg<L, K, 8>, 4>::get("upx_band_kernel<upx::Cfg<" #L ", " #K ", 8>, 4>");
#define UPX_REG_SIZES(K) UPX_REG(8, K) UPX_REG(9, K) UPX_REG(10, K) UPX_REG(11, K) UPX_REG(12, K) UPX_REG(13, K)
        UPX_REG_SIZES(2) UPX_REG_SIZES(4) UPX_REG_SIZES(8)
        return t;
    }();
    auto it = table.find(std::make_tuple(log2n, k));
    return it == table.end() ? nullptr : &it->second;
}
}   // namespace upxk
