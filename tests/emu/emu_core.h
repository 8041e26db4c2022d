#pragma once
#include <functional>
#include <stdint.h>
namespace kxemu {
void launch(uint32_t nblocks, const std::function<void()>& fn);                       // one 64-lane wave per block
void launch_block(uint32_t nblocks, int waves, const std::function<void()>& fn);      // `waves` waves per workgroup
extern int failed;
}
