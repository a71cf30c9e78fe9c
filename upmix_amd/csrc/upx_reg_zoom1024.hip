// upx_reg_zoom1024.hip - band-limited path (upx_zoom.h) with 1024-point sub-transforms: analysis and synthesis kernels for 8 and
// 16 residues per workgroup (decimation 4 is not instantiated: measured slower than the fused kernel).
#include "upx_kernels.h"

namespace upxk {
const ZoomEntry* find_zoom_p1024(int rg, int k) {
    static const std::map<std::tuple<int, int>, ZoomEntry> table = [] {
        std::map<std::tuple<int, int>, ZoomEntry> t;
#define UPX_ZOOM(RG, K)                                                                              \
    t[std::make_tuple(RG, K)] = ZoomImpl<upx::ZoomCfg<10, RG, K>>::get(                             \
        "upx_zoom_analysis_kernel<upx::ZoomCfg<10, " #RG ", " #K ">>",                              \
        "upx_zoom_synthesis_kernel<upx::ZoomCfg<10, " #RG ", " #K ">>");
        UPX_ZOOM(8, 2) UPX_ZOOM(8, 4) UPX_ZOOM(8, 8) UPX_ZOOM(16, 2) UPX_ZOOM(16, 4) UPX_ZOOM(16, 8)
        return t;
    }();
    auto it = table.find(std::make_tuple(rg, k));
    return it == table.end() ? nullptr : &it->second;
}
}   // namespace upxk
