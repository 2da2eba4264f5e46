// hooks.hip.h -- the only places where a measurement harness may reach into a product kernel.  In the library
// build every hook below is empty / the default; a harness under tools/ defines the macro BEFORE it includes a
// translation unit of csrc/ (e.g. `#define BPG_TL(slot) my_stamp(slot)`), so no experiment code lives in csrc/.
#ifndef BPG_HOOKS_HIP_H
#define BPG_HOOKS_HIP_H

// k_rollout: time stamps at a few points of the launch (slot = a small integer)
#ifndef BPG_TL
#define BPG_TL(slot) do { } while (0)
#endif
// k_step_record: entry (0) / exit (1) stamp of a block; agp = the achieved_goal slot the launch writes
#ifndef BPG_TL_STEP
#define BPG_TL_STEP(slot, agp) do { } while (0)
#endif
// blocks per CU the fused rollout's register budget is sized for
#ifndef BPG_ROLLOUT_MINB
#define BPG_ROLLOUT_MINB 1
#endif
// rollout_geometry(): the batch size from which the fused rollout runs in 1024-lane blocks, and the lanes' worth of blocks per CU a
// grid is capped at (A/B experiments of the launch shape)
#ifndef BPG_ROLLOUT_FAT_FROM
#define BPG_ROLLOUT_FAT_FROM 524288
#endif
#ifndef BPG_ROLLOUT_CU_LANES
#define BPG_ROLLOUT_CU_LANES 1024
#endif
// write-once outputs of the one-shot kernels leave as nontemporal stores (state.hip.h, OUT_NT)
#ifndef BPG_OUT_NT
#define BPG_OUT_NT 1
#endif

#endif
