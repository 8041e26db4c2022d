#pragma once
#include <functional>
#include <stdint.h>
namespace kxemu {
void launch(uint32_t nblocks, const std::function<void()>& fn);   // one 64-lane wave per block
extern int failed;
}
