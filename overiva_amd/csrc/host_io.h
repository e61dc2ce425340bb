// Host-side helpers of the epilogue's hand-over to the caller (host_io.hip).
#pragma once
#include <cstddef>

#include "oiva_internal.h"

namespace oiva {

constexpr int kHostRingSlots = 3;
// the process-wide ring of pinned staging buffers, each of at least `bytes` (grown on demand, freed at exit)
hipError_t host_ring_slots(size_t bytes, void** slots);
// dst[r][0 .. row_bytes) = src[r][0 .. row_bytes) for r < nrows, by the pool's threads (the caller is one of them)
void host_copy_rows(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t row_bytes, long long nrows);
// fault the pages of [ptr, ptr + bytes) in for writing, contents kept, by the pool's threads (MADV_POPULATE_WRITE; where the
// kernel lacks it and may_touch, a read-modify-write of one byte per page -- only for ranges nobody else writes meanwhile)
void host_prefault(void* ptr, size_t bytes, bool may_touch);
int host_io_threads();

}  // namespace oiva
