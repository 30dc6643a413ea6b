"""Every scheduling / routing knob of the product path in ONE place.

The knobs are read from `TECOGAN_*` environment variables, parsed ONCE into a `Tuning` object (`current()`); the engines and
the step take their settings from that object when they are built - nothing else in the package reads the environment (the
library path `TECOGAN_LIB` of `_lib.py`, needed before anything else can load, is the one exception and is listed here too).
`current()` re-parses only when one of the listed variables changed since the last parse (tests and the A/B tools set them
between two constructions); objects already built keep what they were built with.

Each knob carries its default and the file under `profiles/` (or the test) that measured it: INTEGRATION.md's table is
generated from `KNOBS` (`python -m pytorch_tecogan_amd.tuning`), and tests/test_host_cpu.py holds the defaults to the
documented optimum.  Variants that were built, measured slower and rejected live behind `-DTG_EXPERIMENTS`
(`csrc/build.sh --experiments`): their knobs are marked `experiment=True` and raise unless that library is loaded."""
import os
from collections import OrderedDict, namedtuple

Knob = namedtuple("Knob", "env attr kind default evidence doc experiment")


def _k(env, attr, kind, default, evidence, doc, experiment=False):
    return Knob("TECOGAN_" + env, attr, kind, default, evidence, doc, experiment)


# kind: "on" (anything but "0" is on), "flag" (only "1" is on), "int", "str", "oint" / "ostr" (None when unset: "the package decides")
KNOBS = OrderedDict((k.attr, k) for k in (
    # ---- step schedule (step.TecoGANStep)
    _k("GRAPH", "graph", "on", True, "profiles/r01_h_bench_6p1ms.json",
       "replay the step as per-lane hipGraphs (0: eager launches)"),
    _k("LANES", "lanes", "on", True, "profiles/r02_a_overlap_probe_priority_cumask.log, r02_d_schedule_matrix.log",
       "two lanes of per-stream graphs (0: one forked capture on one stream)"),
    _k("CU_RESERVE", "cu_reserve", "int", 0, "profiles/r02_b_lane_matrix.log, r02_m_cu_reserve_rerun.log",
       "CUs masked off the real half's stream for the chain (0: none; every value measured slower)"),
    _k("DREAL_BWD", "dreal_bwd", "on", True, "profiles/r02_d_schedule_matrix.log",
       "the real half's backward beside the chain (0: batched with the fake half's)"),
    _k("DP_INLINE", "dp_inline", "on", True, "profiles/r03_q_dp_inline.log",
       "data parallel: one synchronous all-reduce per network on its lane's stream (0: asynchronous, see DP_BUCKETS)"),
    _k("DP_BUCKETS", "dp_buckets", "on", True, "profiles/r03_j_bucket_ends_rccl_1rank.log",
       "with DP_INLINE=0: two buckets per network issued where they become final (0: one asynchronous all-reduce per network)"),
    _k("FORCE_COLLECTIVES", "force_collectives", "flag", False, "tests/test_step_gpu.py (one-rank RCCL group)",
       "issue the collectives even with one rank (test hook)"),
    # ---- persistent launches: workgroup caps (kernels.persist_wgs*)
    _k("PERSIST_WGS", "persist_wgs", "oint", None, "profiles/r02_q_persist_wgs_sweep.log",
       "cap of every persistent launch (unset: G 160, D 96; for steps of <= 4096 LR pixels per pass G 144, its forward launches 160)"),
    _k("PERSIST_WGS_G", "persist_wgs_g", "oint", None, "profiles/r03_r_rw_dma_ab.log", "generator's cap (unset: 144 for steps of <= 4096 LR pixels per pass, else 160)"),
    _k("PERSIST_WGS_D", "persist_wgs_d", "oint", None, "profiles/r02_q_persist_wgs_sweep.log", "discriminator's cap (unset: 96)"),
    _k("PERSIST_WGS_DREAL", "persist_wgs_dreal", "oint", None, "profiles/r05_l_caps_resweep.log",
       "cap of the discriminator's REAL half, which runs beside the chain (unset: the D cap; 72 / 80 for chain-bound steps until the end of round 5)"),
    _k("PERSIST_FWD_G", "persist_fwd_g", "int", 0, "profiles/r04_z_fwd_cap.log, r05_v_caps_write_through.log",
       "cap of the generator's FORWARD register-weights launches (the chain, beside the real half); 0: 160 for steps of <= 4096 LR pixels per pass, else the generator's"),
    _k("PERSIST_TRUNK_G", "persist_trunk_g", "int", 0, "profiles/r04_z_trunk_cap.log, r06_h_caps.log",
       "cap of the trunk's 32 input-gradient launches in the batched G backward (0: 160 for steps of <= 4096 LR pixels per pass - two tiles "
       "per workgroup instead of 2.2 at the generator's 144 -, else the generator's)"),
    # (round 5 pruned five knobs that every sweep of rounds 2-4 had left at their defaults: PERSIST_RW_G / _D - a separate cap for the
    #  register-weights launches, r02_q -, PERSIST_FWD_DREAL / _DFAKE - r04_z_d_fwd_caps - and RW_DHALF_OFF - r03_r)
    # ---- kernel routing (engine.Conv, kernels.rw_eligible)
    _k("RW", "rw", "str", "1", "profiles/r02_c_mb_rw.log", "register-weights 3x3 kernel: 0 never, 1 where measured faster, all"),
    _k("RW_EXTRA", "rw_extra", "str", "trunk,c30,m128,s3,s1", "profiles/r03_r_rw_dma_ab.log, r04_l_rw_extra_s3.log, r04_x_rw_fwd_routing.log",
       "launch classes routed to it for the STEP's sake (capped persistent launches are better neighbours)"),
    _k("RW_EXTRA_DREAL", "rw_extra_dreal", "ostr", None, "profiles/r03_r_rw_dma_ab.log",
       "more classes for the discriminator's real half only (unset: s1 for chain-bound steps)"),
    _k("RW_FWD_MIN", "rw_fwd_min", "int", 4096, "profiles/r04_x_rw_fwd_routing.log",
       "FORWARD 3x3 launches (Cin 64 / 128, any Cout multiple of 64) of at least this many pixels go to it too (0: off): "
       "config 2 3.92 -> 3.80 ms (16384) -> 3.79 (4096), config 4 9.7 -> 9.4-9.5 ms, config 5 2840 -> 2946 frames/s"),
    _k("PAIR_RW_MIN", "pair_rw_min", "int", 32768, "profiles/r04_x_rw_fwd_routing.log, r06_u_knobs.log",
       "conv_trans.2's conv pair as two register-weights launches instead of the fused block launch from this many pixels (0: never): "
       "config 5 3126 -> 3208 frames/s; round 6: 16384 -> 32768, i.e. config 2's chain (4 x 64 x 64 pixels) back on the ONE fused launch "
       "(3.237 -> 3.22 ms: a launch less on the critical path of each of the ten passes), configs[3] / [4] unchanged"),
    _k("INFER_CHUNK", "infer_chunk", "int", 16, "profiles/r04_z_inference_chunks.log",
       "frames per hipGraph (and per staging copy) in RecurrentGenerator"),
    _k("INFER_WGS", "infer_wgs", "int", 256, "profiles/r04_x_rw_fwd_routing.log",
       "workgroup cap of the generator's persistent launches inside RecurrentGenerator (no other lane to leave CUs to)"),
    _k("SUBPIX_CT", "subpix_ct", "on", True, "profiles/r01_h_bench_6p1ms.json (tools/mb_convt.py)",
       "conv-transpose forward as one four-class sub-pixel launch"),
    _k("C3_CW", "c3_cw", "on", True, "profiles/r05_z_conv3_cw_ab.log",
       "register-weights 3x3 launches with 64 reduction channels on the eight-equal-waves kernel (conv3_cw.hip; 0: conv3_rw.hip's producer / consumer form)"),
    _k("C4D_CW", "c4d_cw", "on", True, "profiles/r05_zz_conv4s2d_cw_ab.log",
       "input-gradients of the 4x4 stride-2 convolutions (64 / 128 reduction channels, 16-bit) on the persistent class-waves kernel (conv4s2d_cw.hip; 0: the sub-pixel launch)"),
    _k("CT_CW", "ct_cw", "on", True, "profiles/r05_r_convt_cw_ab.log",
       "conv-transpose FORWARD launches (Cin 64 / 128, 16-bit) on the persistent class-waves kernel (convt_cw.hip; 0: the sub-pixel / four-class launches)"),
    _k("S2_CW", "s2_cw", "on", True, "profiles/r06_b_conv_s2_cw_ab.log",
       "4x4 stride-2 FORWARD launches and the conv-transposes' input-gradients (64 / 128 reduction channels, 16-bit) on the persistent "
       "register-weights kernel (conv_s2_cw.hip; 0: conv4s2_mfma.hip's per-tile staging)"),
    _k("FAST_C4S2", "fast_c4s2", "on", True, "profiles/r01_h_bench_6p1ms.json (tools/mb_c4s2.py)",
       "dedicated stride-2 kernels (4x4 s2 forward / input-gradient, conv-transpose input-gradient)"),
    _k("RGB_OUT", "rgb_out", "on", True, "profiles/r02_i_bench_5p25ms.json", "output layer forward on conv_rgb.hip"),
    _k("RGB_BWD", "rgb_bwd", "on", True, "profiles/r03_o_rgb_bwd_ab.log", "output layer backward as one launch (rgb_bwd.hip)"),
    _k("RGB_BWD_WGS", "rgb_bwd_wgs", "int", 0, "profiles/r03_o_rgb_bwd_ab.log, r06_u_knobs.log",
       "its workgroups (0: 160 up to 40 x 128 x 128 output pixels - the launch runs beside the fake half's 96 -, 256 beyond)"),
    _k("FUSED_RESBLOCK", "fused_resblock", "on", True, "profiles/r01_h_bench_6p1ms.json",
       "conv-relu-conv(+skip) of the trunk in one launch (resblock.hip)"),
    _k("RB_WS", "rb_ws", "on", True, "profiles/r05_a_resblock_ws_ab.log",
       "the fused trunk block as the wave-specialised, stream-first kernel (resblock_ws.hip: LDS-DMA patch and W1, 32x32x16 tiles "
       "without a split-K exchange, W2 in registers; 0: resblock.hip)"),
    _k("RB_PREFETCH", "rb_prefetch", "flag", False, "profiles/r04_p_resblock_prefetch_ab.log (chain 1.97 -> 1.86 ms, step 4.01 -> 3.90 ms without)",
       "a fused trunk block touches the NEXT block's weights (L2 prefetch hint of tg_resblock_fwd); paid in round 2, costs now"),
    _k("FUSED_RESBLOCK_BWD", "fused_resblock_bwd", "flag", False, "DESIGN.md 'resblock' (15 x 28.2 vs 30 x 14.9 us: equal)",
       "both input-gradients of a trunk block in one launch"),
    # ---- weight gradients
    _k("WGRAD_LIST", "wgrad_list", "on", True, "profiles/r03_a_mb_wgrad_group.log, r03_a_wgrad_group_step_ab.log",
       "work-list weight-gradient launches (wgrad_group.hip)"),
    _k("WGRAD_GROUPS", "wgrad_groups", "on", True, "profiles/r01_f_bench_8p45ms.json", "same-shape layers in one grid (fp32 mode)"),
    _k("DEFER_FINALIZE", "defer_finalize", "on", True, "profiles/r01_f_bench_8p45ms.json", "one slab fold per network and backward part"),
    _k("FOLD_ITEMS", "fold_items", "on", True, "profiles/r03_d_fold_items_ab.log", "item-indexed fold (0: blocks_per_job x jobs grid)"),
    _k("PACK_BLOCKS", "pack_blocks", "int", 48, "commit 0a9ce4b (16 / 48 / 96: 4.186 / 4.177 / 4.182 ms)",
       "workgroups per job of the weight repack launch"),
    # ---- batch norm
    _k("STATS_REPLICAS", "stats_replicas", "int", 4, "profiles/r02_k_mb_stats_replicas.log, r02_k_stats_replicas_ab.log",
       "replica blocks of the batch-norm accumulators (a power of two)"),
    _k("D_TAIL", "d_tail", "on", True, "profiles/r06_d_tail_ab.log",
       "the discriminator's tail (BN + LeakyReLU of block4, block5, fc, sigmoid [, the real half's loss seed]; and its backward down to "
       "block4's output gradient) as ONE single-workgroup launch per direction (d_tail.hip; 0: 5-6 + 6 separate launches)"),
    # ---- element type / library / launcher
    _k("DTYPE", "dtype", "str", "bf16", "BASELINE.json configs[1]", "compute element type when args.tg_dtype is unset (bf16 | fp16 | fp32)"),
    _k("LIB", "lib", "ostr", None, "tools/ab_libs.sh", "path of an alternative libtecogan_hip.so (A/B builds); read by _lib.py at import"),
    _k("DIST_BACKEND", "dist_backend", "str", "nccl", "tests/test_dp_gpu.py", "main.py's process-group backend (gloo: ranks sharing one GPU, tests)"),
    # ---- rejected experiments: need the -DTG_EXPERIMENTS library (csrc/build.sh --experiments)
    _k("RB_PAIR", "rb_pair", "flag", False, "profiles/r03_t_resblock2_ab.log (chain 1.56 -> 1.71 ms)",
       "two trunk blocks per launch, halo recomputed (resblock2.hip)", True),
    _k("RB_PAIR_WS", "rb_pair_ws", "flag", False, "profiles/r05_b_resblock2_ws_ab.log (6.1 vs 5.05 us per block)",
       "two trunk blocks per launch of the stream-first kernel (resblock2_ws.hip, halo recomputed)", True),
    _k("BN_FUSE", "bn_fuse", "flag", False, "profiles/r03_g_bn_fuse_ab.log (step 4.43 -> 4.46 ms)",
       "batch-norm backward sums in the producing input-gradient's epilogue (stats_mode 3)", True),
    _k("BN_BWD_FUSED", "bn_bwd_fused", "flag", False, "profiles/r03_l_bn_bwd_fused_ab.log (step 4.405 -> 4.42 ms)",
       "single-launch batch-norm backward for small tensors (tg_bn_bwd_fused)", True),
    _k("BN_BWD_COOP", "bn_bwd_coop", "flag", False, "profiles/r04_y_bn_bwd_coop_ab.log (step 3.77 -> 4.52 ms)",
       "batch-norm backward as ONE cooperative launch (sums, grid-wide wait, apply; tg_bn_bwd_coop)", True),
    _k("MASK_BITS", "mask_bits", "flag", False, "profiles/r06_k_mask_bits_ab.log, r06_n_mask_code_ab.log (G backward alone 1.262 -> 1.251 ms, chain alone 1.215 -> 1.236: step +0.02-0.03 ms)",
       "conv_trans.4's forward also writes the 1-bit mask of its ReLU output and c6's input-gradient (the batched G backward's largest "
       "launch) reads that instead of the 168-MB activation; bit-for-bit the same results - and slower: four more store instructions per "
       "tile in the CHAIN cost more than the input-gradient gains", True),
    _k("WGRAD_B128_PIXELS", "wgrad_b128_pixels", "int", 0, "profiles/r03_m_wgrad_b128.log (1.1-2.5x slower: spills)",
       "64 x 128 channel blocks in the work lists for layers with at least this many pixels (0: never)", True),
))

_SIG = None
_CUR = None


class Tuning:
    """parsed knob values as attributes (see KNOBS); `explicit` = the set of attrs whose variable was set"""

    def __init__(self, env=None):
        env = os.environ if env is None else env
        self.explicit = set()
        for k in KNOBS.values():
            raw = env.get(k.env)
            if raw is not None:
                self.explicit.add(k.attr)
            if k.kind == "on":
                v = k.default if raw is None else raw != "0"
            elif k.kind == "flag":
                v = k.default if raw is None else raw == "1"
            elif k.kind == "int":
                v = k.default if raw is None else int(raw)
            elif k.kind == "oint":
                v = None if raw is None else int(raw)
            else:  # "str" / "ostr"
                v = k.default if raw is None else raw
            setattr(self, k.attr, v)
        r = self.stats_replicas
        if r < 1 or r & (r - 1):
            raise ValueError("TECOGAN_STATS_REPLICAS must be a power of two")

    # ---- derived values (the defaults' history is in kernels.py next to the functions that use them)
    def cap(self, net):
        """workgroups of network `net`'s persistent launches ('G', 'D' or None = the global cap)"""
        base = self.persist_wgs if self.persist_wgs is not None else 160
        if net == "G":
            return self.persist_wgs_g if self.persist_wgs_g is not None else base
        if net == "D":
            if self.persist_wgs_d is not None:
                return self.persist_wgs_d
            return base if self.persist_wgs is not None else min(base, 96)
        return base

    def cap_g_for(self, lr_pixels):
        """the generator's cap for a training step of `lr_pixels` = B * h * w, or None when the environment fixes it"""
        if self.persist_wgs is not None or self.persist_wgs_g is not None:
            return None
        # chain-bound steps (<= 4096 LR pixels per pass): 144 since the chain's forward convolutions are persistent launches too
        # (3.79 -> 3.75 ms with the rest of r04_x's routing; 128: 3.80); larger steps keep 160 (config 4: 9.72 vs 9.5-9.6 ms)
        return 144 if lr_pixels <= 4096 else self.cap(None)

    def cap_fwd_g_for(self, lr_pixels):
        """cap of the generator's FORWARD register-weights launches for such a step (0: the generator's cap): the chain's launches
        run beside the real half (96 workgroups; 72 / 80 earlier) and may take more of the chip than the backward pass beside the fake half's 96 -
        176 / 192 / 208 / 256: 3.727 / 3.735 / 3.763 / 3.751 vs 3.762-3.780 ms at the generator's 144 (profiles/r04_z_fwd_cap.log).
        Round 5, with the register-weights kernels' results written through the L2: 128 / 144 / 160 / 176 / 192 / 224 = 3.41-3.42 / 3.41 /
        3.404-3.406 / 3.42-3.44 / 3.44 / 3.44-3.45 ms (profiles/r05_v_caps_write_through.log)"""
        if self.persist_fwd_g:
            return self.persist_fwd_g
        return 160 if lr_pixels <= 4096 else 0

    def rgb_bwd_wgs_for(self, hr_pixels):
        """workgroups of the output layer's one-pass backward for a batch of `hr_pixels` = N * H * W output pixels"""
        if self.rgb_bwd_wgs:
            return self.rgb_bwd_wgs
        return 160 if hr_pixels <= 40 * 128 * 128 else 256

    def cap_trunk_g_for(self, lr_pixels):
        """cap of the trunk's 32 input-gradient launches of the batched G backward (0: the generator's).  Round 4 found no effect
        (107 ... 256: 3.724-3.737 ms, lane B was the long pole then); since the discriminator's tail became two launches lane A ends
        last, and 40 x 32 x 32 pixels = 320 tiles are TWO per workgroup at 160 instead of 3 / 2 at 144: 0 / 160 / 176 / 192 / 256 =
        3.268 / 3.234 / 3.241 / 3.245 / 3.240 ms (profiles/r06_h_caps.log)"""
        if self.persist_trunk_g:
            return self.persist_trunk_g
        return 160 if lr_pixels <= 4096 else 0

    def cap_dreal_for(self, lr_pixels):
        """cap of the discriminator's REAL half for such a step, or None (environment fixes it / step is not chain-bound)"""
        if self.persist_wgs is not None or self.persist_wgs_d is not None or self.persist_wgs_dreal is not None:
            return None
        # 72 through round 4; 80 since the chain got 0.2 ms shorter (round 5: the real half, not the chain, bounds phase 1 now):
        # 64 / 72 / 80 / 88 / 96 / 112: 3.56-3.57 / 3.54-3.56 / 3.485-3.50 / 3.50-3.51 / 3.555 / 3.57 ms (profiles/r05_l_caps_resweep.log).
        # End of round 5 (write-through stores, the stride-2 input-gradients persistent at this cap): 80 / 88 / 96 / 104 / 112 / 128 =
        # 3.37 / 3.34-3.36 / 3.31-3.32 / 3.33-3.34 / 3.34 / 3.38 ms (profiles/r05_v_caps_write_through.log) - the discriminator's own cap
        return 96 if lr_pixels <= 4096 else None

    def cap_dreal_default(self):
        return self.persist_wgs_dreal if self.persist_wgs_dreal is not None else self.cap("D")


def _signature():
    g = os.environ.get
    return tuple(g(k.env) for k in KNOBS.values())


def current():
    """the Tuning object for the environment as it is now (cached; one parse per distinct setting)"""
    global _SIG, _CUR
    sig = _signature()
    if _CUR is None or sig != _SIG:
        _CUR, _SIG = Tuning(), sig
    return _CUR


def need_experiments(attr):
    """called where a knob marked `experiment` is switched on: the rejected variants are not in the default library"""
    from . import _lib as L
    if not L.has_experiments():
        raise L.TecoganHipError(f"{KNOBS[attr].env} needs the experiments build of libtecogan_hip.so "
                                "(pytorch-tecogan_amd/csrc/build.sh --experiments; default builds leave the rejected variants out)")


def markdown_table():
    rows = ["| Variable | Default | What it does | Evidence |", "|---|---|---|---|"]
    for k in KNOBS.values():
        d = {True: "1", False: "0", None: "(unset)"}.get(k.default, k.default) if not isinstance(k.default, int) or isinstance(k.default, bool) \
            else k.default
        rows.append(f"| `{k.env}`{' (experiments build)' if k.experiment else ''} | `{d}` | {k.doc} | {k.evidence} |")
    return "\n".join(rows)


if __name__ == "__main__":
    print(markdown_table())
