// upx_reg_big.hip - unfused pipeline (upx_big.h): STFT sizes 64 .. 65536 with any hop.
#include "upx_kernels.h"

namespace upxk {
const BigEntry* find_big(int log2n) {
    static const std::map<int, BigEntry> table = [] {
        std::map<int, BigEntry> t;
#define UPX_BIG(L) t[L] = BigImpl<upx::BigCfg<L>>::get();
        UPX_BIG(6) UPX_BIG(7) UPX_BIG(8) UPX_BIG(9) UPX_BIG(10) UPX_BIG(11) UPX_BIG(12) UPX_BIG(13) UPX_BIG(14) UPX_BIG(15) UPX_BIG(16)
#undef UPX_BIG
        return t;
    }();
    auto it = table.find(log2n);
    return it == table.end() ? nullptr : &it->second;
}
}   // namespace upxk
