// log2 of the LDS tile of one NTT pass (shared by the kernel and the host code that sequences the passes)
#pragma once
namespace masp {
static constexpr int NTT_LT = 10;
}  // namespace masp
