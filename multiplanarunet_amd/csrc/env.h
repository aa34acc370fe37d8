// Every environment switch of the library in ONE table: name, kind, default, what it does. Values are read once per process,
// at the first query (tests that flip a switch run a fresh interpreter); mpu_env_describe() prints the table with the values
// in force, and tests/test_unet_host.py checks that no other getenv lives in csrc/ and that DESIGN.md names every switch.
// Switches are schedule / fusion on-off pairs for A/B runs and dev aids -- the 18 tuning thresholds of rounds 1-2 became
// constants in round 3.
#pragma once

namespace mpu {

enum EnvKind {
    ENV_ON,        // on unless the variable starts with '0'
    ENV_OFF,       // off unless the variable starts with '1'
    ENV_NUM,       // integer (atol), the default when unset
    ENV_IMPL       // MPU_CONV_IMPL: "regs" = 0 (register-staged reference kernels), anything else = 1
};

// X(id, "NAME", kind, default, "what it does")
#define MPU_ENV_TABLE(X)                                                                                                              \
    X(CONV_IMPL, "MPU_CONV_IMPL", ENV_IMPL, 1, "regs: the register-staged round-1 conv / wgrad kernels everywhere (reference schedules)")       \
    X(CONV_HALO, "MPU_CONV_HALO", ENV_ON, 1, "0: no LDS-resident-patch kernels (conv_c8 / conv_ws / conv_halo*): everything on conv_glds / conv_pipe") \
    X(CONV_C8, "MPU_CONV_C8", ENV_ON, 1, "0: first layer (8 padded channels) not on conv_c8")                                        \
    X(CONV_WS, "MPU_CONV_WS", ENV_ON, 1, "0: 64-channel level-0 layers not on the weight-stationary conv_ws")                        \
    X(CONV_PIPE, "MPU_CONV_PIPE", ENV_ON, 1, "0: deep layers on conv_glds instead of the 8-wave split-K conv_pipe")                  \
    X(CONV_DEEPK, "MPU_CONV_DEEPK", ENV_ON, 1, "0: no conv_deepk (3x3 on 16-pixel maps with K split over the waves of a workgroup, no split-K partials)") \
    X(PIPE_DEBUG, "MPU_PIPE_DEBUG", ENV_NUM, 0, "dev aid: 32 = s_memtime stamps in conv_pipe")                                       \
    X(HALO8, "MPU_HALO8", ENV_ON, 1, "0: 192-400-workgroup grids on the 4-wave conv_halo instead of the 8-wave conv_halo8")          \
    X(HALO8_SCHED, "MPU_HALO8_SCHED", ENV_NUM, 1, "0: conv_halo8 with lockstep halves (round-3 A/B); 1: halves one phase apart")      \
    X(XCD_TILES, "MPU_XCD_TILES", ENV_ON, 1, "0: conv_halo / conv_halo8 / conv_ws tiles dealt to workgroups in launch order instead of XCD-contiguous ranges (round-6 A/B)") \
    X(HALO_UPCONV, "MPU_HALO_UPCONV", ENV_ON, 1, "0: up-convolutions not on the low-resolution-patch conv_halo variant")             \
    X(HALO_UP8_MIN, "MPU_HALO_UP8_MIN", ENV_NUM, 2048, "grid size from which up-convolutions take 8-row tiles")                      \
    X(HALO_KNOCKOUT, "MPU_HALO_KNOCKOUT", ENV_NUM, 0, "dev aid (-DMPU_HALO_KNOCKOUT_BUILD only): knock-out mask of the predict conv kernel") \
    X(HALO16P, "MPU_HALO16P", ENV_ON, 1, "0: large inference grids not on the persistent conv_halo16p (round-3 schedules instead)")  \
    X(HALO16_MIN, "MPU_HALO16_MIN", ENV_NUM, -1, "grid bound of conv_halo16p in tiles (default 512; tests: 1)")   \
    X(HALO16P_WGS, "MPU_HALO16P_WGS", ENV_NUM, 0, "cap on conv_halo16p's persistent workgroups (0 = one per CU; tests: few, many tiles each)") \
    X(FUSED_HEAD, "MPU_FUSED_HEAD", ENV_ON, 1, "0: inference 1x1 head as its own kernel instead of the last conv's epilogue")        \
    X(FUSED_POOL, "MPU_FUSED_POOL", ENV_ON, 1, "0: inference 2x2 max pooling as its own kernel instead of a second epilogue output") \
    X(FUSED_BN_STATS, "MPU_FUSED_BN_STATS", ENV_ON, 1, "0: BatchNorm statistics by colreduce instead of the conv epilogue")          \
    X(BN_FOLD, "MPU_BN_FOLD", ENV_ON, 1, "0: BatchNorm finalize as its own launch also where the producer left <= 64 partial rows (round-5 A/B)") \
    X(BN_ATOMIC, "MPU_BN_ATOMIC", ENV_ON, 1, "0: the fused BatchNorm sums as partial rows + finalize launches everywhere (round-5 form) instead of fixed-point accumulators") \
    X(FUSED_BN_BWD_CONV, "MPU_FUSED_BN_BWD_CONV", ENV_ON, 1, "0: BatchNorm-backward sums by colreduce instead of the data-gradient epilogue") \
    X(FUSED_BN_BWD, "MPU_FUSED_BN_BWD", ENV_ON, 1, "0: max-pool backward + skip add without the fused BatchNorm-backward sums")      \
    X(HEAD_TRAIN_FUSED, "MPU_HEAD_TRAIN_FUSED", ENV_ON, 1, "0: training step with the post-BatchNorm tensor of the last block materialised (BN apply, head forward, head backward, column reduction, BN backward as five launches) instead of the three head_bn_* passes (round-6 A/B)") \
    X(POOL_BWD_RECOMPUTE, "MPU_POOL_BWD_RECOMPUTE", ENV_NUM, 1, "0: encoder levels' backward step with the post-BatchNorm tensor read and the summed gradient (skip + un-pooled) written between max-pool backward and BatchNorm backward, instead of both passes recomputing them (round-6 A/B); 1: recompute from 4 M elements per level; 2: at every level (tests)") \
    X(POOL_BWD_BLOCKS, "MPU_POOL_BWD_BLOCKS", ENV_NUM, 0, "dev aid: cap on the workgroups of the two pool-backward recompute passes (0 = the default)") \
    X(HEAD_RS, "MPU_HEAD_RS", ENV_ON, 1, "0: head forward without the reduce-scatter variant")                                       \
    X(WGRAD_C8, "MPU_WGRAD_C8", ENV_ON, 1, "0: first-layer weight gradient not on wgrad_c8")                                         \
    X(WGRAD_TAPS, "MPU_WGRAD_TAPS", ENV_ON, 1, "0: no wgrad_taps (strip-resident weight gradients): wgrad_glds everywhere")          \
    X(WGRAD_TAPS_STAG, "MPU_WGRAD_TAPS_STAG", ENV_ON, 1, "0: wgrad_taps with lockstep wave groups (round-3 A/B)")                    \
    X(WGRAD_GROUP, "MPU_WGRAD_GROUP", ENV_ON, 1, "0: every weight-gradient kernel as its own launch instead of the grouped launches") \
    X(WGRAD_BATCHED_REDUCE, "MPU_WGRAD_BATCHED_REDUCE", ENV_ON, 1, "0: the K-split reduction right behind every weight-gradient kernel") \
    X(TAIL_OVERLAP, "MPU_TAIL_OVERLAP", ENV_ON, 1, "0: mpu_unet_backward_adam runs the optimizer behind the weight gradients instead of beside them (round-6 A/B)") \
    X(GEOM_FAST, "MPU_GEOM_FAST", ENV_ON, 1, "0: geometry kernels on the op-by-op fp64 path only (no screened fast path)")           \
    X(FUSE_FX, "MPU_FUSE_FX", ENV_ON, 1, "0: fused back-mapping with fp64 index arithmetic instead of the fixed-point screen")       \
    X(PROF_MARKERS, "MPU_PROF_MARKERS", ENV_OFF, 0, "1: roofline-leg events as hipEventRecord markers instead of dispatch-bound events") \
    X(STAMPS, "MPU_STAMPS", ENV_OFF, 0, "dev aid: 1 = the s_memtime-instrumented kernel instantiations + stamp buffer")

enum EnvId {
#define MPU_ENV_ID(id, name, kind, dflt, doc) ENV_##id,
    MPU_ENV_TABLE(MPU_ENV_ID)
#undef MPU_ENV_ID
    ENV_COUNT
};

long env(EnvId id);                              // the switch's value (read from the environment once per process)

}  // namespace mpu
