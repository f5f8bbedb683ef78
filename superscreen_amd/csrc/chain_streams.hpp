// The high-priority streams that the panel chains of the factorization schedules (chol.hip, lu.hip) run on, and
// the measurement that decides which of them a schedule uses.
#pragma once
#include "common.hpp"

namespace ssa {

constexpr int kChainPool = 16;

// The chain streams of the current device in the order in which a schedule whose chip-filling launches (trailing
// updates) run on `load` should use them: out[0 .. count) (count <= kChainPool; beyond the number of distinct good
// streams the list wraps around).  Measured on first use per device and load stream (about 10 ms), then cached.
// The streams belong to the library (ssa_shutdown destroys them); a schedule must join everything it put on them
// into its caller's stream before it returns.
int chain_streams_get(hipStream_t load, int count, hipStream_t *out);

// Diagnostics: the cost of a dependent launch (microseconds) and the pipe group (0 = the load stream's own pipe) of
// up to `capacity` chain streams of the current device, in order of use, from the most recent measurement.
// Returns the number of chain streams, 0 if nothing has been measured on this device.
int chain_streams_costs(double *microseconds, int32_t *pipe_group, int capacity);

// Forgets the measurements of the current device; the next schedule measures again.
int chain_streams_invalidate();

int chain_streams_shutdown();

}  // namespace ssa
