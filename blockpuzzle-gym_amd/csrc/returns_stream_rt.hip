// returns_stream_rt.hip -- the register-resident return kernel at horizons other than 50 / 25 / 10: rows t >= T are neither read
// nor do they contribute, only the first T entries of a row leave (k_discounted_return_lane<32 | 64, ..., RT = true>).
#include "returns_lane.hip.h"

int bpgk::launch_return_lane_rt(const float *d_reward_tb, int64_t n, int64_t ld, int T, double gamma, double *d_G, hipStream_t st)
{
    const PowTable tab = pow_table(gamma, T);
    const int plain_ok = gamma_is_subtractable(gamma, T) ? 1 : 0;
    const dim3 pgrid((unsigned)((n + 64 * DR_LANE_WPB - 1) / (64 * DR_LANE_WPB))), pblock(64 * DR_LANE_WPB);
    if (T <= 32)
        hipLaunchKernelGGL((k_discounted_return_lane<32, DRM_EXACT, DR_LANE_WPB, G_TILE_ROWS, true>), pgrid, pblock, 0, st, tab,
                           d_reward_tb, n, ld, gamma, d_G, plain_ok, T);
    else
        hipLaunchKernelGGL((k_discounted_return_lane<64, DRM_EXACT, DR_LANE_WPB, G_TILE_ROWS, true>), pgrid, pblock, 0, st, tab,
                           d_reward_tb, n, ld, gamma, d_G, plain_ok, T);
    return check_launch("k_discounted_return_lane<rt>");
}
