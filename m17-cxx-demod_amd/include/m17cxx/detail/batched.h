// The device handle the batched overloads of the operator classes take: one C-ABI context (include/m17hip.h) = one GPU.
// The scalar forms of the classes need none of this; a host that only uses those never touches libm17hip.so.
#pragma once

#include "../../../../include/m17hip.h"

#include <cstdint>
#include <stdexcept>
#include <string>

namespace mobilinkd
{
namespace batched
{

class Device
{
    m17hip_ctx* ctx_ = nullptr;

public:
    // room for `max_channels` channels x `max_samples` samples per call
    Device(uint32_t max_channels, uint32_t max_samples, int device = 0)
    {
        const int r = m17hip_ctx_create(device, max_channels, max_samples, &ctx_);
        if (r != M17HIP_OK) throw std::runtime_error(std::string("m17hip_ctx_create: ") + m17hip_strerror(r));
    }
    ~Device() { m17hip_ctx_destroy(ctx_); }
    Device(const Device&) = delete;
    Device& operator=(const Device&) = delete;
    m17hip_ctx* ctx() const { return ctx_; }
};

} // batched
} // mobilinkd
