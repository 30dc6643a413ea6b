/*
 * tecogan_hip.h - C ABI of libtecogan_hip.so (gfx950 / MI355X).
 *
 * The reference (dwight-foster/Pytorch-TecoGAN) has no FFI: its hot path is Python calling ATen ops.
 * Every entry point below therefore replaces an ATen call site of the reference; the citation after each
 * declaration names that call site (file:line are into the reference tree).  INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions:
 *   - arguments are raw device pointers, int32 dims, a dtype enum and a hipStream_t (passed as void*);
 *   - the caller owns every buffer including workspaces; functions only enqueue work on `stream`
 *     (no allocation, no synchronisation, no global state beyond one-time function-attribute setup);
 *   - return value: 0 ok, <0 invalid argument / unsupported shape (TG_E_*), >0 a hipError_t;
 *   - nothing throws across the ABI; distinct streams may be used from distinct threads.
 *
 * Device tensor layout: activations are NHWC with the channel count padded to a multiple of 32
 * ("Cp"); element type is TG_F32 or TG_BF16.  Weights are consumed in a packed, fragment-ordered
 * layout produced by tg_pack_conv_weights from the PyTorch-layout fp32 master copy.
 */
#ifndef TECOGAN_HIP_H
#define TECOGAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TG_ABI_VERSION 4   /* 4 (round 6): + tg_conv4s2_fwd_cw, tg_convt_dgrad_cw, tg_d_tail_fwd / _bwd / _max_pixels / _scratch_floats, TG_MASK_RELU_BITS + tg_convt_fwd_cw's relu_bits; 3 (round 5): + tg_convt_fwd_cw, tg_conv3x3_cw, tg_conv4s2_dgrad_cw, tg_conv4s2_fwd_capped; 2 (round 5): + tg_resblock_fwd_ws; round 4 removed tg_wgrad_group / tg_absdiff_nchw and moved the
                           * rejected variants behind TG_EXPERIMENTS without a bump */

enum { TG_F32 = 0, TG_BF16 = 1, TG_F16 = 2 };  /* TG_F16: IEEE half, same layouts as TG_BF16 (loss scaling: tg_adam) */

enum {
  TG_OK = 0,
  TG_E_BADARG = -1,      /* null pointer / non-positive dim / bad enum */
  TG_E_UNSUPPORTED = -2, /* shape outside what the kernels are instantiated for */
  TG_E_ALIGN = -3        /* channel count not a multiple of 32, pointer not 16-byte aligned */
};

enum { TG_ACT_NONE = 0, TG_ACT_RELU = 1, TG_ACT_LRELU = 2, TG_ACT_SIGMOID = 3, TG_ACT_TANH24 = 4 /* 24*tanh, code/models.py:50 */ };
enum { TG_MASK_NONE = 0, TG_MASK_RELU = 1, TG_MASK_LRELU = 2, TG_MASK_BNZ = 3 /* tg_conv stats_mode 3: `mask` is z (experiments build only) */,
       TG_MASK_RELU_BITS = 4 /* tg_conv3x3_rw only: `mask` is the 1-BIT form of a ReLU output, [N][H][W][Cout / 8] bytes, bit c % 8 of byte
                              * c / 8 = (stored value of channel c > 0), as tg_convt_fwd_cw writes it (relu_bits): a sixteenth of the bytes
                              * a masked input-gradient tile reads back (round 6) */ };
enum { TG_OUT_NHWC = 0, TG_OUT_NCHW_F32 = 1 };

#define TG_MAX_TAPS 16
#define TG_MAX_CLASSES 4

/* One output sub-lattice ("class") of a gather convolution.  For every class-grid pixel (cy,cx) the kernel
 * computes   acc[co] = sum_t sum_ci in[n, cy*S + dy[t], cx*S + dx[t], ci] * W[widx[t]][co][ci]
 * (out-of-range input pixels read as zero) and stores it at out[n, cy*OS + ooy, cx*OS + oox, co]. */
typedef struct {
  int32_t ooy, oox;
  int32_t ntaps;
  int8_t dy[TG_MAX_TAPS];
  int8_t dx[TG_MAX_TAPS];
  int16_t widx[TG_MAX_TAPS];
} tg_conv_class;

typedef struct {
  int32_t dtype;                /* TG_F32 / TG_BF16: element type of in/out/res/mask and of the packed weights */
  int32_t N, IH, IW, Cin;       /* input  [N][IH][IW][Cin]   (Cin multiple of 32) */
  int32_t OH, OW, Cout;         /* output [N][OH][OW][Cout]  (Cout multiple of 32) */
  int32_t S, OS;                /* input step per class-grid pixel; output step */
  int32_t ncls;
  tg_conv_class cls[TG_MAX_CLASSES];
  /* epilogue, applied in this order: +bias, +res, act, *mask, store, per-channel stats */
  int32_t act;                  /* TG_ACT_* */
  int32_t mask_mode;            /* TG_MASK_*: multiply by act'(mask) where mask is the saved activation */
  int32_t stats_mode;           /* 0 none, 1 per-channel sum, 2 sum and sum of squares, 3 (with TG_MASK_BNZ) sum out and sum out * z:
                                 * the batch-norm backward sums of the layer whose output gradient this launch produces */
  int32_t stats_groups;         /* batch is split into this many equal groups with separate statistics */
  int32_t out_mode;             /* TG_OUT_* */
  int32_t c_real;               /* TG_OUT_NCHW_F32: number of real channels stored (<=4) */
  int64_t out_n_stride;         /* TG_OUT_NCHW_F32: element stride between samples of `out` */
  int32_t tile_cfg;             /* 0 auto; otherwise a TG_TILE_* id (tuning / tests) */
  int32_t stats_replicas;       /* 0/1: statistics go to stats[groups][2][Cout]; R (power of two): spread over
                                   stats[R][groups][2][Cout] by pixel-tile index, fold with tg_reduce_replicas */
} tg_conv_desc;

enum { TG_TILE_AUTO = 0, TG_TILE_64x256 = 1, TG_TILE_64x64 = 2, TG_TILE_128x128 = 3, TG_TILE_32x128 = 4,
       TG_TILE_32x64 = 5, TG_TILE_64x128 = 6,
       TG_TILE_64x128_8W = 7, TG_TILE_64x64_8W = 8 /* 8 waves (two per SIMD); plain 3x3 launches only, else the 4-wave tile */ };
/* <output channels>x<pixels> per workgroup */

int tg_abi_version(void);
const char* tg_error_string(int code);
/* 1 when the library was built with -DTG_EXPERIMENTS (csrc/build.sh --experiments): it then also exports the entry points declared
 * under `#ifdef TG_EXPERIMENTS` below and accepts the arguments marked "experiments build" - variants that were built, measured
 * slower and rejected, kept buildable so that the A/B logs under profiles/ can be reproduced.  The default library is
 * what the default step can reach. */
int tg_has_experiments(void);

/* Bytes needed for the packed weights of a conv with `nslots` weight slots (taps). */
int64_t tg_packed_weight_bytes(int dtype, int nslots, int cout_p, int cin_p);

/* fp32 PyTorch-layout weights -> packed fragment-ordered weights of element type `dtype`.
 *   packed[slot][chunk][row][kc]  with  value = w[co*s_co + ci*s_ci + slot_off[slot]]  (0 where co/ci are padding)
 * slot_off gives, per packed slot, the element offset of that tap inside one (co,ci) kernel (e.g. kh*KW+kw), so the
 * same routine packs forward weights (s_co = Cin*KH*KW, s_ci = KH*KW), transposed-conv weights and the
 * role-swapped copies that the dgrad launches use.  Replaces the implicit weight handling inside
 * aten::conv2d / conv_transpose2d (code/ops.py:45-63). */
int tg_pack_conv_weights(int dtype, const float* w, void* packed, int cout, int cin, int cout_p, int cin_p,
                         int64_t s_co, int64_t s_ci, int nslots, const int32_t* slot_off_dev, void* stream);

/* The same for every conv of a network in ONE launch.  jobs_dev: njobs x 9 int64 on the device =
 * {w ptr, packed ptr, s_row, s_k, rows, K, rows_p, K_p, nslots}; slot t reads kernel offset t. */
int tg_pack_conv_weights_multi(int dtype, const int64_t* jobs_dev, int njobs, int blocks_per_job, void* stream);

/* Gather convolution on MFMA: conv3x3 (code/models.py:54-58,68,73-76,102 via code/ops.py:57-63), conv4x4 stride 2
 * (code/models.py:90-94), conv-transpose k3 s2 p1 op1 as four sub-pixel classes (code/ops.py:45-54;
 * code/models.py:72,74) and the input-gradient of each (aten::convolution_backward, code/train.py:336,340). */
/* Launch plan of tg_conv for this descriptor: TG_TILE_* id (AUTO resolved) | (1<<8 if the compile-time 3x3 variant is
 * used); negative TG_E_* on an invalid descriptor.  For tooling/benchmarks; launches nothing. */
int tg_conv_pick_tile(const tg_conv_desc* d);
int tg_conv(const tg_conv_desc* d, const void* in, const void* w_packed, const float* bias, const void* res,
            const void* mask, void* out, float* stats, void* stream);

/* The same 3x3 stride-1 convolution (flip = 0: forward, taps dy = t/3-1, dx = t%3-1; flip = 1: input-gradient, taps mirrored,
 * w_packed = the role-swapped packing) for DENSE launches: persistent workgroups that keep their weights in registers and
 * walk pixel tiles of 8x16 with a double-buffered LDS patch (csrc/conv3_rw.hip).  bf16, Cin in {64, 128}, Cout % 64 == 0,
 * else TG_E_UNSUPPORTED (use tg_conv).  Epilogue as tg_conv: +bias, +res, act (NONE/RELU/LRELU), *act'(mask), NHWC store,
 * stats (may be null; stats_mode 1: per-channel sums, 2: sums and sums of squares; stats_replicas blocks of
 * [stats_groups][2][Cout], ACCUMULATED - see "replica blocks" at tg_bn_apply).
 * max_workgroups: 0 = one per CU (256); the grid is min(tiles, max_workgroups / (Cout/64)) x Cout/64. */
int tg_conv3x3_rw(int dtype, const void* in, const void* w_packed, const float* bias, const void* res, const void* mask,
                  void* out, float* stats, int N, int H, int W, int Cin, int Cout, int flip, int act, int mask_mode,
                  int stats_mode, int stats_groups, int stats_replicas, int max_workgroups, void* stream);

/* The same operation, arguments and results (up to the fp32 summation order of the statistics) for Cin = 64 with EIGHT EQUAL WAVES
 * (csrc/conv3_cw.hip, round 5): every wave keeps the 9 taps' weights of 32 output channels in registers, multiplies 2 of the tile's
 * 8 rows, brings its share of the next patch by LDS-DMA and finishes its pixels straight from the accumulators - no producer /
 * consumer roles, no accumulator image, one barrier per tile.  TG_E_UNSUPPORTED unless Cin == 64 (tg_conv3x3_rw takes 128). */
int tg_conv3x3_cw(int dtype, const void* in, const void* w_packed, const float* bias, const void* res, const void* mask,
                  void* out, float* stats, int N, int H, int W, int Cin, int Cout, int flip, int act, int mask_mode,
                  int stats_mode, int stats_groups, int stats_replicas, int max_workgroups, void* stream);

/* Conv-transpose k3 s2 p1 op1 forward (code/ops.py:45-54 conv2_tran; code/models.py:72,74) as ONE sub-pixel launch: a
 * workgroup computes all four output classes of its input tile from one staged patch (tg_conv runs the classes as four sets
 * of workgroups).  in [N][IH][IW][Cin] -> out [N][2IH][2IW][Cout], w_packed = the 9-slot forward packing, epilogue
 * +bias, act in {NONE, RELU, LRELU}.  TG_E_UNSUPPORTED unless Cout % 64 == 0 (use tg_conv then). */
int tg_convt_fwd(int dtype, const void* in, const void* w_packed, const float* bias, void* out, int N, int IH, int IW,
                 int Cin, int Cout, int act, void* stream);

/* The same layer (same arguments, same results up to the fp32 summation order) with CLASS-SPECIALISED waves (csrc/convt_cw.hip,
 * round 5): min(tiles, max_workgroups / (Cout/64)) x Cout/64 persistent workgroups walk 4 x 16 INPUT tiles; each of the eight waves
 * keeps the slots of ONE sub-pixel class x 32 of the 64 output channels in registers for the workgroup's lifetime and stores its
 * class's results straight from the accumulators (no accumulator image, one barrier per tile; the patch comes by LDS-DMA).
 * bf16 / fp16, Cin in {64, 128}, Cout % 64 == 0, else TG_E_UNSUPPORTED (use tg_convt_fwd).  max_workgroups: 0 = one per CU. */
int tg_convt_fwd_cw(int dtype, const void* in, const void* w_packed, const float* bias, void* out, int N, int IH, int IW,
                    int Cin, int Cout, int act, void* relu_bits, int max_workgroups, void* stream);
/* (relu_bits, may be null, act must be TG_ACT_RELU: also writes the 1-bit mask of the stored output, [N][2IH][2IW][Cout / 8] bytes - see
 * TG_MASK_RELU_BITS) */

/* 4x4 stride-2 padding-1 conv forward (the discriminator's down-sampling convs, code/models.py:90-94) with compile-time
 * taps and pipelined chunk staging; in [N][IH][IW][Cin] (IH, IW even) -> out [N][IH/2][IW/2][Cout]; w_packed = the 16-slot
 * forward packing; bias may be null; stats (may be null) = stats_replicas blocks of [stats_groups][2][Cout] per-channel sum /
 * sum of squares of the stored output, ACCUMULATED (zero it first; replica blocks: see tg_bn_apply).  TG_E_UNSUPPORTED unless
 * Cout % 64 == 0 (use tg_conv then). */
int tg_conv4s2_fwd(int dtype, const void* in, const void* w_packed, const float* bias, void* out, float* stats,
                   int stats_groups, int stats_replicas, int N, int IH, int IW, int Cin, int Cout, void* stream);

/* The same launch with at most max_workgroups workgroups (rounded up to a multiple of 8; 0: one per (pixel tile, channel tile) unit):
 * a workgroup walks its units.  The step passes the discriminator's cap - no faster alone, 0.015 ms of the step beside the other lane. */
int tg_conv4s2_fwd_capped(int dtype, const void* in, const void* w_packed, const float* bias, void* out, float* stats,
                          int stats_groups, int stats_replicas, int N, int IH, int IW, int Cin, int Cout, int max_workgroups,
                          void* stream);

/* Input-gradient of the 4x4 stride-2 convs (autograd of code/models.py:90-94) as one four-class sub-pixel launch (the
 * tg_convt_fwd kernel with a 3x3 window and 16 (class, tap) pairs): dout [N][OH][OW][Cout] -> din [N][2OH][2OW][Cin];
 * w_dgrad_packed = the role-swapped 16-slot packing; mask (may be null) [N][2OH][2OW][Cin] = saved activation of the layer
 * below, the result is multiplied by act'(mask) (TG_MASK_*).  TG_E_UNSUPPORTED unless Cin % 64 == 0 (use tg_conv then). */
int tg_conv4s2_dgrad(int dtype, const void* dout, const void* w_dgrad_packed, void* din, int N, int OH, int OW, int Cout,
                     int Cin, const void* mask, int mask_mode, void* stream);

/* The same input-gradient (same arguments and results up to the fp32 summation order) with CLASS-SPECIALISED waves
 * (csrc/conv4s2d_cw.hip, round 5; the structure of tg_convt_fwd_cw): persistent workgroups over 4 x 16 tiles of dout, one
 * sub-pixel class x 32 output channels per wave with its 4 slots' weights in registers, act'(mask) and the stores straight
 * from the accumulators.  bf16 / fp16, Cout (the reduction) in {64, 128}, Cin % 64 == 0, else TG_E_UNSUPPORTED.
 * max_workgroups: 0 = one per CU. */
int tg_conv4s2_dgrad_cw(int dtype, const void* dout, const void* w_dgrad_packed, void* din, int N, int OH, int OW, int Cout,
                        int Cin, const void* mask, int mask_mode, int max_workgroups, void* stream);

/* Input-gradient of the conv-transpose layers (autograd of code/ops.py:45-54) with the same kernel structure (3x3-window
 * stride-2 gather): dout [N][OH][OW][Cout] (OH, OW even) -> din [N][OH/2][OW/2][Cin]; w_dgrad_packed = the role-swapped
 * 9-slot packing.  TG_E_UNSUPPORTED unless Cin % 64 == 0 (use tg_conv then). */
int tg_convt_dgrad(int dtype, const void* dout, const void* w_dgrad_packed, void* din, int N, int OH, int OW, int Cout,
                   int Cin, void* stream);

/* The two stride-2 gathers above with the WEIGHTS IN REGISTERS (csrc/conv_s2_cw.hip, round 6; same arguments and results up to the
 * fp32 summation order): persistent workgroups over 4 x 16 output tiles; the reduction (taps x channels) is split between four wave
 * groups whose partial sums meet through an LDS exchange; the patch arrives by LDS-DMA with its columns de-interleaved by parity.
 * tg_conv4s2_fwd_cw replaces aten::conv2d of code/models.py:90-94 (statistics as tg_conv4s2_fwd); tg_convt_dgrad_cw the input
 * gradient of code/ops.py:45-54.  bf16 / fp16, reduction channels (Cin / Cout of the respective signature) in {64, 128}, output
 * channels % 64 == 0, else TG_E_UNSUPPORTED (use the launches above).  max_workgroups: 0 = one per CU. */
int tg_conv4s2_fwd_cw(int dtype, const void* in, const void* w_packed, const float* bias, void* out, float* stats,
                      int stats_groups, int stats_replicas, int N, int IH, int IW, int Cin, int Cout, int max_workgroups,
                      void* stream);
int tg_convt_dgrad_cw(int dtype, const void* dout, const void* w_dgrad_packed, void* din, int N, int OH, int OW, int Cout,
                      int Cin, int max_workgroups, void* stream);

/* Weight gradient: slab[split][t][a][b] = sum over the split's pixels of X[n, y*S+dy[t], x*S+dx[t]][a] * Y[n,y,x][b].
 * (aten::convolution_backward weight path, code/train.py:336,340.) */
typedef struct {
  int32_t dtype;
  int32_t N, XH, XW, Cx;        /* X [N][XH][XW][Cx] */
  int32_t YH, YW, Cy;           /* Y [N][YH][YW][Cy] */
  int32_t S;
  int32_t ntaps;                /* 9 or 16 */
  int8_t dy[TG_MAX_TAPS];
  int8_t dx[TG_MAX_TAPS];
  int32_t nsplit;               /* number of pixel-range splits (= slabs) */
  int32_t taps_per_wg;          /* 0: every workgroup owns all taps; 3 (3x3) / 4 (4x4): taps are split over blockIdx.z */
  int32_t y_sum;                /* 1: every slab also carries sum over pixels of Y[.][b] (Cy floats behind the taps) = the
                                 * bias gradient of a conv whose Y operand is its output gradient */
} tg_wgrad_desc;

int64_t tg_wgrad_slab_floats(const tg_wgrad_desc* d);
int tg_wgrad(const tg_wgrad_desc* d, const void* x, const void* y, float* slab, void* stream);
/* The same launch for `njobs` layers of identical shape (e.g. the 33 64-channel 3x3 layers of the generator trunk, whose
 * backward passes all exist once the dgrad chain is done): jobs_dev = njobs x {x, y, slab} device pointers as int64.  One
 * grid covers every layer, so the chip is filled by layers x splits and each layer needs ~njobs times fewer slabs. */
int tg_wgrad_multi(const tg_wgrad_desc* d, const int64_t* jobs_dev, int njobs, void* stream);
/* grad[a*s_a + b*s_b + slot_off[t]] (+)= sum_split slab[split][t][a][b]  for a<ca, b<cb.  Writes the PyTorch layout.
 * slab_stride = floats per split as tg_wgrad wrote them (tg_wgrad_slab_floats / nsplit).  bias_grad non-null (slabs
 * written with y_sum=1): bias_grad[b] += sum_split of the channel sums. */
int tg_wgrad_finalize(const float* slab, int nsplit, int ntaps, int ca_p, int cb_p, int ca, int cb, float* grad,
                      int64_t s_a, int64_t s_b, const int32_t* slot_off_dev, int accumulate, float* bias_grad,
                      int64_t slab_stride, void* stream);

/* The same for every conv of a network in ONE launch (always accumulates; slot t adds kernel offset t).
 * jobs_dev: njobs x 12 int64 = {slab ptr, grad ptr, s_a, s_b, nsplit, ntaps, ca_p, cb_p, ca, cb, bias-grad ptr or 0,
 * slab stride in floats}. */
int tg_wgrad_finalize_multi(const int64_t* jobs_dev, int njobs, int blocks_per_job, void* stream);
/* The same fold with one workgroup per work item instead of blocks_per_job workgroups per job.  A job's items are its
 * (16 x BB)-channel tiles x chunks of 8 slabs: (ca_p / (1024 / BB)) * (cb_p / BB) * ceil(nsplit / 8), BB = 64 if cb_p % 64 == 0
 * else 32.  jobs_dev: njobs x 13 int64 = the 12 entries above + the job's first item index (prefix sum, ascending);
 * nitems = their total; max_taps = 9 or 16 (the largest ntaps among the jobs: sizes the LDS image). */
int tg_wgrad_fold_items(const int64_t* jobs_dev, int njobs, int nitems, int max_taps, void* stream);

/* ---- grouped weight gradients of 3x3 stride-1 convolutions (16-bit element types; csrc/wgrad_group.hip) -----------------
 * ONE persistent launch for a list of layers of possibly different image size / channel counts (the weight path of
 * aten::convolution_backward behind code/train.py:336,340 for the plain 3x3 layers of code/models.py:54-58,68-76,90-94).
 * Work unit = (job, 64 x 64 channel block, 128-pixel tile of tile_w x 128/tile_w pixels); the units of all jobs form one
 * list (job-major, then block = a_block * b_blocks + b_block, then tile = (n * tiles_y + ty) * tiles_x + tx) and workgroup
 * w of W' = ceil(units_total / per), per = ceil(units_total / workgroups), takes units [w * per, (w + 1) * per).
 * jobs_dev: njobs x 12 int64 = {x ptr, y ptr, first unit of the job, N, H, W, Cx, Cy, tiles_x = ceil(W / tile_w),
 * tiles_y = ceil(H / (128 / tile_w)), y_sum (0/1), ordinal of the job's first channel block among all blocks of the list};
 * x [N,H,W,Cx], y [N,H,W,Cy] NHWC, Cx % 32 == Cy % 32 == 0; channel blocks are 64 x 64, a_blocks = ceil(Cx / 64),
 * b_blocks = ceil(Cy / 64) (a 32-channel remainder is a block whose upper half holds don't-care values: fold real channels only).
 * slab: slots of tg_wgrad_group_slot_floats_v(TG_WGROUP_C3) floats = [9 taps][64 a][64 b] partial dW + [64] channel sums of Y (the conv's
 * bias gradient; zeros unless y_sum and a_block == 0; entries of padded channels are don't-care).  The segment of workgroup w inside channel block g (global ordinal)
 * goes to slot w + g: block g owns slots [w_first(g) + g, w_last(g) + g], at most W' + (number of blocks) slots in all,
 * and tg_wgrad_finalize_multi folds them with one job per block (ca_p = cb_p = 64, stride = the slot size).
 * tile_w: 32 (tiles of 32 x 4 pixels) or 16 (16 x 8).  TG_E_UNSUPPORTED for fp32: use tg_wgrad.
 * Entry point: tg_wgrad_group_v(dtype, TG_WGROUP_C3, ...) below.
 * The same work-list launch for the two stride-2 layer kinds (tile_w = 16: tiles of 16 x 4 pixels of y; H, W of a job row
 * are y's, x lives on the 2H x 2W grid; 64-pixel tiles, so tiles_x = ceil(W / 16), tiles_y = ceil(H / 4)):
 *   TG_WGROUP_CT    conv-transpose k3 s2 p1 op1 (code/ops.py:45-54): x = the output gradient [N,2H,2W,Cx], y = the layer input
 *                   [N,H,W,Cy]; dW[t][a][b] = sum x[n, 2y + dy[t], 2x + dx[t]][a] * y[n, y, x][b], 9 taps (dy, dx) in -1..1;
 *   TG_WGROUP_C4S2  conv k4 s2 p1 (code/models.py:90-94): x = the layer input [N,2H,2W,Cx], y = the output gradient [N,H,W,Cy];
 *                   16 taps (dy, dx) in -1..2.
 * TG_WGROUP_C3 is the 3x3 stride-1 kind described above.  Slot size: tg_wgrad_group_slot_floats_v(variant) = [taps][64][64] + [64].
 *   TG_WGROUP_C3_B128 / TG_WGROUP_CT_B128 (experiments build only, else TG_E_UNSUPPORTED: 1.1-2.5x slower, spills): the same two kinds with 64 x 128 channel blocks (b_blocks = ceil(Cy / 128), slot =
 *                   [9][64][128] + [128]; fold jobs with cb_p = 128): for layers with >= 128 y channels - x is fetched once per
 *                   128 of them and a 32-pixel k-step is 26 transposed LDS reads per 36 MFMAs instead of 22 per 18. */
#define TG_WGROUP_C3 0
#define TG_WGROUP_CT 1
#define TG_WGROUP_C4S2 2
#define TG_WGROUP_C3_B128 3
#define TG_WGROUP_CT_B128 4
int64_t tg_wgrad_group_slot_floats_v(int variant);
int tg_wgrad_group_v(int dtype, int variant, int tile_w, const int64_t* jobs_dev, int njobs, int units_total, int workgroups,
                     float* slab, void* stream);

/* ---- output layer of the generator (code/models.py:77-79 conv 64 -> 3 + sigmoid; the store replaces the permute + float()
 * of code/train.py:97-99) --------------------------------------------------------------------------------------------
 * out[n * out_n_stride + c * H * W + y * W + x] = act(conv3x3(in, w)[n][y][x][c] + bias[c]) for c < c_real <= 4, fp32;
 * in NHWC [N][H][W][64] (16-bit), w_packed = the forward packing of tg_pack_conv_weights for Cout padded to 32 (9 slots);
 * act = TG_ACT_NONE or TG_ACT_SIGMOID.  Same result as tg_conv with TG_OUT_NCHW_F32 (one 16-row MFMA tile instead of two).
 * TG_E_UNSUPPORTED for fp32 or Cin != 64: use tg_conv then. */
int tg_conv3x3_rgb(int dtype, const void* in, const void* w_packed, const float* bias, float* out, long long out_n_stride,
                   int c_real, int N, int H, int W, int Cin, int act, void* stream);

/* Backward of that layer for the batched generator backward (autograd of code/models.py:77-79 behind code/train.py:336), one pass:
 *   dx[n][y][x][ci] = (x > 0) * sum_{t, co} dpre4[p - off(t)][co] * w[co][ci][t]      (input gradient under the ReLU mask of x)
 *   slab slot of workgroup g: partial dW in tg_wgrad's slab layout [9 taps][64 ci][32] fp32, columns co < 3 (column 3 zero, the
 *   rest unwritten): fold with ONE job {slab, gw, s_a = 9, s_b = 576, nsplit = workgroups, 9, 64, 32, 64, 3, 0, slot}.
 * dpre4 [N][H][W][4] 16-bit (tg_content_loss with dpre_channels = 4), x / dx [N][H][W][64], w = the fp32 master weight
 * [3][64][3][3].  workgroups = min(N * ceil(H / 16) * ceil(W / 16), max_workgroups) persistent workgroups; slab must hold that
 * many slots of tg_conv3x3_rgb_bwd_slot_floats() floats.  TG_E_UNSUPPORTED for fp32 or Cin != 64 (tg_conv + tg_wgrad then). */
int64_t tg_conv3x3_rgb_bwd_slot_floats(void);
int tg_conv3x3_rgb_bwd(int dtype, const void* dpre4, const void* x, const float* w, void* dx, float* slab, int N, int H, int W,
                       int Cin, int max_workgroups, void* stream);

/* ---- fused residual block (code/ops.py:45-54; code/models.py:66-69), bf16, C == 64 ------------------------------
 * out_h = relu(conv3x3(in, w1) + b1), out_a = (add_skip ? in : 0) + conv3x3(out_h, w2) in ONE launch (add_skip = 0: the
 * conv-relu-conv pair of conv_trans.2, code/models.py:73); tensors NHWC [N][H][W][64]; w1 / w2 are
 * the forward packings of tg_pack_conv_weights (9 slots).  out_h is rounded to bf16 before the second conv, exactly like
 * the two-launch sequence.  out_h may be null: the intermediate activation is then not stored (inference: only the backward pass reads
 * it).  TG_E_UNSUPPORTED for any other dtype / channel count (run two tg_conv launches then).
 * next_w1_packed / next_w2_packed (both or neither; may be null): packed weights of the residual block that will be
 * launched next; the kernel touches them so that they sit in L2 when that launch streams them (values are not used). */
int tg_resblock_fwd(int dtype, const void* in, const void* w1_packed, const float* b1, const void* w2_packed, void* out_h,
                    void* out_a, int N, int H, int W, int C, int add_skip, const void* next_w1_packed,
                    const void* next_w2_packed, void* stream);

/* The same block, same arguments and results up to the fp32 summation order (conv1's K is ONE accumulator chain here, two added
 * halves there), as the wave-specialised, stream-first kernel of round 5 (csrc/resblock_ws.hip): patch and W1 by LDS-DMA from
 * tick 0, conv1 as v_mfma_f32_32x32x16 tiles on four compute waves without a split-K exchange, W2 straight into registers.
 * The default of the recurrent pass and of inference (TECOGAN_RB_WS=0: tg_resblock_fwd). */
int tg_resblock_fwd_ws(int dtype, const void* in, const void* w1_packed, const float* b1, const void* w2_packed, void* out_h,
                       void* out_a, int N, int H, int W, int C, int add_skip, void* stream);

#ifdef TG_EXPERIMENTS
/* The same two input-gradients as ONE PERSISTENT, tile-pipelined launch (csrc/resblock_pp.hip, round 5): min(tiles, max_workgroups)
 * workgroups keep both weight sets for all their 8 x 4 tiles (W of stage 1 in an LDS image for 32x32x16 tiles, W of stage 2 in
 * registers) and stream only the 12 x 8 patch and the relu mask per tile; the four conv1 waves work on tile i while the four conv2
 * waves finish tile i - 1.  For launches with many tiles per workgroup (the batched generator backward: 40 x 32 x 32 = 1280 tiles).
 * MEASURED SLOWER than two register-weights launches (22.5 vs 20.5 us per block at 40 x 32 x 32, 55 vs 41 at 32 x 64 x 64, 144 workgroups:
 * ~4300 ticks per 8 x 4 tile of which ~1150 are matrix work - profiles/r05_h_resblock_pp_ab.log): experiments build only.
 * max_workgroups: 0 = one per CU (256).  Same operands, results (up to the fp32 summation order) and limits as tg_resblock_bwd. */
int tg_resblock_bwd_pp(int dtype, const void* dout, const void* w2_dgrad_packed, const void* h, const void* w1_dgrad_packed,
                       void* out_dh, void* out_din, int N, int H, int W, int C, int max_workgroups, void* stream);

/* TWO consecutive residual blocks in ONE launch (8 x 4 output tiles, halo recomputed: h1 on 14 x 10, a1 on 12 x 8, h2 on 10 x 6
 * pixels from a 16 x 12 patch): out_h1 = relu(conv(in, w1a) + b1a), out_a1 = in + conv(out_h1, w2a), out_h2 = relu(conv(out_a1, w1b)
 * + b1b), out_a2 = out_a1 + conv(out_h2, w2b) - bit-identical to two tg_resblock_fwd launches, one launch boundary and one
 * patch round trip fewer on the recurrent pass's serial chain.  next_w4: host array of the NEXT launch's four packed weight
 * images (L2 prefetch hint) or null.  Same dtype / channel limits as tg_resblock_fwd. */
int tg_resblock2_fwd(int dtype, const void* in, const void* w1a_packed, const float* b1a, const void* w2a_packed,
                     const void* w1b_packed, const float* b1b, const void* w2b_packed, void* out_h1, void* out_a1, void* out_h2,
                     void* out_a2, int N, int H, int W, int C, const void* const* next_w4, void* stream);

/* TWO consecutive residual blocks in ONE launch of the same stream-first structure (csrc/resblock2_ws.hip; 8 x 4 output tiles, halo
 * recomputed: h1 on 14 x 10, a1 on 12 x 8, h2 on 10 x 6 pixels from a 16 x 12 patch): out_h1 = relu(conv(in, w1a) + b1a), out_a1 = in +
 * conv(out_h1, w2a), out_h2 = relu(conv(out_a1, w1b) + b1b), out_a2 = out_a1 + conv(out_h2, w2b) - bit-identical to two
 * tg_resblock_fwd_ws launches; one launch boundary and one start-up of the weight stream fewer per pair on the recurrent pass's serial
 * chain - in theory: measured 6.1 us per block against 5.05 for tg_resblock_fwd_ws (profiles/r05_b_resblock2_ws_ab.log), so it
 * is an experiments-build entry point.  out_h1 / out_h2 may be null (inference).  Meant for launches of up to 128 8 x 8 tiles (the shapes tg_resblock_fwd_ws runs
 * on 8 x 4 tiles); correct for any size.  Same dtype / channel limits. */
int tg_resblock2_fwd_ws(int dtype, const void* in, const void* w1a_packed, const float* b1a, const void* w2a_packed,
                        const void* w1b_packed, const float* b1b, const void* w2b_packed, void* out_h1, void* out_a1, void* out_h2,
                        void* out_a2, int N, int H, int W, int C, void* stream);
#endif

/* Input-gradient of the same block in ONE launch (aten::convolution_backward x2 + threshold_backward, code/train.py:336):
 * out_dh = (h > 0) * conv3x3^T(dout, w2), out_din = dout + conv3x3^T(out_dh, w1); w*_dgrad_packed are the role-swapped
 * ("dgrad") packings of tg_pack_conv_weights; h is the forward pass's out_h.  out_dh is the Y operand of the first conv's
 * weight gradient (and, summed per channel, its bias gradient).  next_*: L2 prefetch hint as in tg_resblock_fwd. */
int tg_resblock_bwd(int dtype, const void* dout, const void* w2_dgrad_packed, const void* h, const void* w1_dgrad_packed,
                    void* out_dh, void* out_din, int N, int H, int W, int C, const void* next_wa_packed,
                    const void* next_wb_packed, void* stream);


/* ---- layout converters ------------------------------------------------------------------------------- */
/* NCHW fp32 (strided samples) -> NHWC `dtype` with zero channel padding. */
int tg_nchw_to_nhwc(int dtype, const float* src, int64_t src_n_stride, void* dst, int N, int C, int Cp, int H, int W,
                    void* stream);
int tg_nhwc_to_nchw(int dtype, const void* src, float* dst, int64_t dst_n_stride, int N, int C, int Cp, int H, int W,
                    void* stream);

/* ---- f_net resampling (code/models.py:9-24: nn.MaxPool2d(2) / nn.Upsample(scale_factor=2, bilinear)) --------- */
/* NHWC `dtype`, C % 32 == 0.  maxpool2: [N][H][W][C] -> [N][H/2][W/2][C] (H, W even).  up2_bilinear: -> [N][2H][2W][C],
 * align_corners=False. */
int tg_maxpool2(int dtype, const void* src, void* dst, int N, int H, int W, int C, void* stream);
int tg_up2_bilinear(int dtype, const void* src, void* dst, int N, int H, int W, int C, void* stream);

/* ---- flow / warp / packing (code/train.py:71-111,138-198; code/ops.py:98-100) -------------------------- */
/* dst_plane[i] = post_a * bilinear_x4(pre * src_plane[i]) + post_b ; planes are h*w (src) and 4h*4w (dst) fp32,
 * addressed by element offsets (nn.Upsample(scale_factor=4, bilinear, align_corners=False)). */
int tg_up4_planes(const float* src, const int64_t* src_off_dev, float* dst, const int64_t* dst_off_dev, int nplanes,
                  int h, int w, float pre, float post_a, float post_b, void* stream);
/* F.grid_sample(bilinear, zeros, align_corners=False) on NCHW fp32 images with a grid that is a REINTERPRETED
 * contiguous (2,H,W) block per sample (code/train.py:84,96,157).  grid_off/img_off are per-sample element offsets.
 * fp16_grid!=0 rounds grid values to fp16 first (code/train.py:98,187).  out may be null; corner_idx (int32
 * [N][H][W][2] = x0,y0) may be null; if sq_ref is non-null accumulates sum((sq_ref - warp)^2) into loss_acc[0]. */
int tg_warp_nchw(const float* img, const int64_t* img_off_dev, const float* grid, const int64_t* grid_off_dev,
                 float* out, int32_t* corner_idx, const float* sq_ref, const int64_t* sq_off_dev, float* loss_acc,
                 int N, int C, int IH, int IW, int GH, int GW, int fp16_grid, void* stream);
/* Generator input [B][h][w][64] (NHWC dtype): ch0-2 = lr frame, ch3-50 = pixel_unshuffle4((warp(prev,grid)+1)/2),
 * rest 0.  prev==null gives the first-frame input (zeros).  (code/train.py:86-88,95-107; main.py:191-213) */
int tg_gen_input(int dtype, const float* lr, int64_t lr_n_stride, const float* prev, int64_t prev_n_stride,
                 const float* grid, int64_t grid_n_stride, void* dst, int B, int h, int w, void* stream);
/* Discriminator input [2*tb][H][H][32]: real rows then fake rows (code/train.py:160-198).  half<0: both halves into
 * dst; half=0/1: only the real/fake half, dst pointing at that half's first row. */
int tg_d_assemble(int dtype, const float* x, const float* y, const float* gen, const float* tvel, void* dst, int B,
                  int T, int K, int h, int border, int half, void* stream);
/* dst[dst_off[i] + e] = src_off[i] < 0 ? 0 : src[src_off[i] + e], e < len: assembles T_vel (code/train.py:147-158)
 * from the pseudo-flow blocks, zero blocks and (with tg_up4_planes) the "back" flow planes. */
int tg_copy_blocks(const float* src, const int64_t* src_off_dev, float* dst, const int64_t* dst_off_dev, int nblocks,
                   int64_t len, void* stream);

/* ---- batch norm, training mode, eps/momentum as code/ops.py:75-77 -------------------------------------- */
/* Replica blocks.  A per-channel accumulator ([groups][2][C] floats) that many workgroups add into is kept as R blocks of
 * that shape (R a power of two, zeroed by the caller): producer workgroup b adds into block b mod R, the consumer sums the
 * blocks in order.  A few hundred workgroups adding into the same 128 floats serialise in L2: the discriminator's stage-1
 * conv takes 10.3 us without statistics, 19.9 with R = 1, 13.0 with R = 4 (tools/mb_stats.py).
 * y = act(gamma*(z-mean)*invstd+beta) (+skip).  stats = stats_replicas blocks of [groups][2][C] sums from the producing conv
 * (tg_conv: tg_conv_desc.stats_replicas).  Block 0 also updates running_mean/var (group after group), adds `groups` to
 * *num_batches_tracked (int64, may be null) and writes mean/invstd to save[groups][2][C]. */
int tg_bn_apply(int dtype, const void* z, const float* stats, int stats_replicas, const float* gamma, const float* beta,
                const void* skip, void* y, float* running_mean, float* running_var, float* save, int N, int HW, int C,
                int groups, int act, float eps, float momentum, int64_t* num_batches_tracked, void* stream);
/* red (red_replicas blocks of [groups][2][C]) += (sum dyp, sum dyp*xhat) where dyp = dy * act'(yact). */
int tg_bn_bwd_reduce(int dtype, const void* dy, const void* yact, const void* z, const float* save, float* red,
                     int red_replicas, int N, int HW, int C, int groups, int act, void* stream);
/* dz = gamma*invstd*(dyp - mean(dyp) - xhat*mean(dyp*xhat)); block 0 accumulates dgamma/dbeta.
 * red_raw = 1: red holds (sum dy, sum dy*z) as the epilogue of the tg_conv launch that PRODUCED dy left them (stats_mode 3,
 * TG_MASK_BNZ; act must be TG_ACT_NONE then) instead of tg_bn_bwd_reduce's (sum dyp, sum dyp*xhat): converted here. */
int tg_bn_bwd_apply(int dtype, const void* dy, const void* yact, const void* z, const float* save, const float* red,
                    int red_replicas, const float* gamma, void* dz, float* dgamma, float* dbeta, int N, int HW, int C,
                    int groups, int act, int red_raw, void* stream);
#ifdef TG_EXPERIMENTS
/* tg_bn_bwd_reduce + tg_bn_bwd_apply as ONE cooperative launch (torch.nn.BatchNorm2d backward behind code/models.py:96-113 of the
 * reference): a thread keeps its pixels' dy / z / yact vectors in registers across a grid-wide wait on the sums, so the tensors are
 * read once.  barrier: one zeroed 32-bit word per call (counts arrived workgroups).  All workgroups must be co-resident, so the
 * call returns TG_E_UNSUPPORTED when the tensor needs more than tg_bn_bwd_coop_max_workgroups() of them (4 or 8 pixels rows of
 * 256 / (C / vector) pixels per workgroup) - callers then take the two launches.  red must be zeroed like for tg_bn_bwd_reduce;
 * it holds the same sums afterwards.  A wait longer than 50 ms writes NaNs instead of hanging.
 * Measured SLOWER than the two launches, alone (+6.7 us per layer) and in the step (3.77 -> 4.52 ms: the launch's workgroups wait
 * for CUs behind the other lane's persistent kernels), profiles/r04_y_bn_bwd_coop_ab.log. */
int tg_bn_bwd_coop_max_workgroups(void);
int tg_bn_bwd_coop(int dtype, const void* dy, const void* yact, const void* z, const float* save, float* red, int red_replicas,
                   const float* gamma, void* dz, float* dgamma, float* dbeta, int N, int HW, int C, int groups, int act,
                   unsigned* barrier, void* stream);
/* The same backward pass (reduce + apply) as ONE launch for a tensor of at most tg_bn_bwd_fused_max_pixels() pixels per group
 * (the discriminator's 16x16 ... 4x4 layers): one workgroup per 16-byte channel piece keeps its share of the tensors in registers
 * between the sums and dz and is the only writer of its channels' dgamma / dbeta (+=).  act: TG_ACT_NONE or TG_ACT_LRELU.
 * TG_E_UNSUPPORTED above the pixel limit (callers take the two-launch path). */
int tg_bn_bwd_fused(int dtype, const void* dy, const void* yact, const void* z, const float* save, const float* gamma,
                    void* dz, float* dgamma, float* dbeta, int N, int HW, int C, int groups, int act, void* stream);
int tg_bn_bwd_fused_max_pixels(void);
#endif

/* ---- heads and losses (code/models.py:143-145; code/train.py:205-333) ----------------------------------- */
int tg_fc_head_fwd(int dtype, const void* feat, const float* w, const float* b, float* prob, int N, int HW, int C,
                   int Cp, void* stream);
int tg_fc_head_bwd(int dtype, const void* feat, const float* w, const float* dlogit, void* dfeat, float* dw,
                   float* db, int N, int HW, int C, int Cp, void* stream);

/* The discriminator's TAIL in ONE launch per direction (csrc/d_tail.hip, round 6: one workgroup per sample; what couples the samples -
 * block5.1's batch statistics, block4.1's backward sums - goes through `scratch` and a ticket, the workgroup that draws the last one
 * finishes for all; nobody waits, no co-residency assumption; replaces, behind block4's
 * convolution, the launches tg_bn_apply, tg_conv (block5.0), tg_bn_apply, tg_fc_head_fwd [, tg_dlogit_real] and tg_fc_head_bwd,
 * 2 x (tg_bn_bwd_reduce, tg_bn_bwd_apply), tg_conv (block5.0's input-gradient) - /root/reference/code/models.py:119-123,137-146,
 * code/ops.py:75-77,85-88, autograd of code/train.py:304-307 - with the same rounding points: every tensor another launch reads is
 * written as before).  Tensors NHWC: z4 n4 dn4 dz4 [N][H4][H4][C4], z5 n5 dz5 [N][H4/2][H4/2][Cp5] (C5 real channels, padding written
 * as zeros); stats4 = the replica blocks block4's convolution accumulated; w5 = block5.0.weight, the fp32 MASTER [C5][C4][4][4]
 * (rounded to the element type in the kernel, as the packer does); fc_w [C5 * (H4/2)^2], channel-major as torch's flatten; save4 /
 * save5 [groups][2][C] (mean, invstd); running statistics / num_batches_tracked may be null (no update).  BatchNorm groups are
 * walked one after the other.  tg_d_tail_bwd: seed_real != 0 (groups must be 1) computes d(loss)/d(logit) of the real half from prob
 * as tg_dlogit_real does and leaves it in dlogit_in_out; parameter gradients are ACCUMULATED (+=); dn4 is scratch.
 * scratch: tg_d_tail_scratch_floats(N, H4, groups) floats, ZERO before the first launch (every launch leaves it zero); one buffer
 * per stream of launches (forward and backward of one discriminator call may share it).
 * TG_E_UNSUPPORTED beyond tg_d_tail_max_pixels() block4 pixels per group, C4 != 64, C5 > 4 (run the separate launches). */
long long tg_d_tail_max_pixels(void);
long long tg_d_tail_scratch_floats(int N, int H4, int groups);
int tg_d_tail_fwd(int dtype, const void* z4, const float* stats4, int stats_replicas, const float* gamma4, const float* beta4,
                  float* rmean4, float* rvar4, int64_t* nbt4, float* save4, void* n4, const float* w5, void* z5,
                  const float* gamma5, const float* beta5, float* rmean5, float* rvar5, int64_t* nbt5, float* save5, void* n5,
                  const float* fc_w, const float* fc_b, float* prob, int N, int H4, int C4, int C5, int Cp5, int groups,
                  float eps, float momentum, float* scratch, void* stream);
int tg_d_tail_bwd(int dtype, const float* dlogit_in_out, const float* prob, const float* cfg, const float* loss_scale,
                  int seed_real, const void* n5, const void* z5, const float* save5, const float* gamma5, const float* fc_w,
                  const float* w5, const void* n4, const void* z4, const float* save4, const float* gamma4, void* dz5, void* dn4,
                  void* dz4, float* g_fc_w, float* g_fc_b, float* dgamma5, float* dbeta5, float* dgamma4, float* dbeta4, int N,
                  int H4, int C4, int C5, int Cp5, int groups, float* scratch, void* stream);
/* acc[0] += sum |a-b| over real channels (layer loss, code/train.py:219-220). */
int tg_absdiff_sum(int dtype, const void* a, const void* b, float* acc, int64_t npix, int C, int Cp, void* stream);
/* The same for njobs tensor pairs in one launch (the four D layer losses of code/train.py:205-226): jobs_dev = njobs x 6 int64
 * {a ptr, b ptr, acc ptr, npix, C, Cp}; blocks_per_job workgroups walk each pair. */
int tg_absdiff_sum_multi(int dtype, const int64_t* jobs_dev, int njobs, int blocks_per_job, void* stream);
/* content loss partial sum and d(pre-sigmoid) for the generator output (code/train.py:239-241):
 * acc (>= 16 floats): acc[0] += sum (gen-y)^2 ; dpre[nhwc] = gscale * 2*(gen-y) * gen*(1-gen) ; bias_acc[c] += sum dpre[c], c < 3
 * (the output layer's bias gradient; bias_acc null: acc + 8).  gen/y are NCHW fp32 (B,T,3,H,W);
 * dpre is NHWC [(t1-t0)*B][H][W][dpre_channels] in (t,b) order and covers frames t0 <= t < t1 only; dpre_channels = 32 (the
 * padded operand of tg_conv / tg_wgrad) or, 16-bit types only, 4 (3 + 1 pad: the compact operand of tg_conv3x3_rgb_bwd).
 * pp_T > 0 (ping-pong, T == 2*pp_T-1): acc[6] += sum |gen_t - gen_{2(pp_T-1)-t}| over t < pp_T-1 and the gradient
 * pp_coef*sign(gen_t - gen_partner) is added before the sigmoid derivative (code/train.py:275-283). */
int tg_content_loss(int dtype, const float* gen, const float* y, void* dpre, float* acc, int B, int T, int H, int W,
                    float gscale, int t0, int t1, int pp_T, float pp_coef, const float* loss_scale, float* bias_acc,
                    int dpre_channels, void* stream);
/* All step scalars on device + d(logit) for the discriminator loss (code/train.py:287-333). */
int tg_loss_finalize(const float* prob, const float* acc, float* scalars, float* dlogit, int tb, const float* cfg,
                     const float* loss_scale, void* stream);

/* d(logit) of the REAL half only: dlogit[n] = -(1/tb) * pr*(1-pr)/(pr+eps), n < tb (the real-half term of t_discrim_loss,
 * code/train.py:304-307; eps = cfg[6]).  Lets the real half's D backward start before the fake half exists;
 * tg_loss_finalize later writes the same values again. */
int tg_dlogit_real(const float* prob, float* dlogit, int tb, const float* cfg, const float* loss_scale, void* stream);

/* dst[i] (+)= sum_r src[r*stride + i], i < n: folds the replicated per-channel statistics of tg_conv. */
int tg_reduce_replicas(const float* src, int replicas, int stride, int n, float* dst, int accumulate, void* stream);

/* ---- optimiser (torch.optim.Adam as built at main.py:239-243) ------------------------------------------- */
/* hyper_dev (8 device floats): lr, beta1, beta2, eps, 1-beta1^t, 1-beta2^t, grad_scale, t (the caller's step count) - in
 * memory so that a captured hipGraph picks up each step's values. */
int tg_adam(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper_dev, void* stream);

/* ---- dynamic loss scaling of the fp16 mode (torch.cuda.amp.GradScaler as used at code/train.py:9,335-342) ---------- */
/* `loss_scale` (device float, nullable) of tg_content_loss / tg_loss_finalize / tg_dlogit_real / tg_cosine_loss multiplies
 * every backward seed, so all gradient tensors and both flat gradient buffers carry the scale.  scaler_state (device
 * floats, 8): 0 scale, 1 growth tracker, 2 found_inf of the generator, 3 of the discriminator, 4 1/scale, 5 / 6 updates of
 * the generator / discriminator skipped so far (tg_scaler_update counts them; optimizer.step() is not called on overflow,
 * code/train.py:337,341, so torch's Adam step count does not advance there: tg_adam_scaled uses t - skipped).
 * tg_check_finite: *flag = 1 if any g[i] is inf/NaN (flag = scaler_state + 2 + which).
 * tg_adam_scaled: tg_adam on g / scale, skipped entirely when found_inf[which] is set (GradScaler.step); bias corrections
 *   for step hyper[7] - scaler_state[5 + which] (hyper[4], hyper[5] as given while nothing has been skipped).
 * tg_scaler_update: the two GradScaler.update() calls of one training step (generator's first), then clears the flags. */
int tg_check_finite(const float* g, int64_t n, float* flag, void* stream);
int tg_adam_scaled(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper_dev,
                   const float* scaler_state, int which, void* stream);
int tg_scaler_update(float* scaler_state, float growth, float backoff, int interval, void* stream);

/* ---- FNet training, opt-in (the reference defines f_net, code/models.py:22-50, and leaves its optimiser commented out:
 * main.py:231,244-245, code/train.py:343-346; DESIGN.md "FNet training" fixes the semantics; parity unpinned) -------------
 * tg_warp_grid_grad: the LR warp loss of code/train.py:78-84,247-249 with the estimator's output as the sampling grid, and its
 *   gradient w.r.t. that grid (aten::grid_sampler_2d_backward, bilinear / zeros / align_corners=False, grid part):
 *   v = sample of img block n (C planes of IH x IW at img_off[n]) at grid (the (2,GH,GW) block at grid_off[n] read as (GH,GW,2));
 *   *loss_acc (nullable) += sum (ref - v)^2;  dgrid block n (same layout as the grid block, at dgrid_off[n]) =
 *   coef [* *loss_scale] * d/dgrid sum_c (ref - v)^2.  All buffers fp32.
 * tg_tanh24_bwd: dpre[n][y][x][c] = dout[n][c][y][x] * (24 - out[n][c][y][x]^2 / 24), c < 2 (d/dp of 24 tanh(p)); NHWC, 32 channels.
 * tg_up2_bilinear_bwd: backward of tg_up2_bilinear (nn.Upsample(scale_factor=2, bilinear)): ddst [N][2H][2W][C] ->
 *   dsrc [N][H][W][C], multiplied by LeakyReLU(0.2)'(lrelu_mask) when lrelu_mask [N][H][W][C] is given. */
int tg_warp_grid_grad(const float* img, const int64_t* img_off_dev, const float* grid, const int64_t* grid_off_dev,
                      const float* ref, const int64_t* ref_off_dev, float* dgrid, const int64_t* dgrid_off_dev, float* loss_acc,
                      int N, int C, int IH, int IW, int GH, int GW, float coef, const float* loss_scale, void* stream);
int tg_tanh24_bwd(int dtype, const float* dout_nchw, const float* out_nchw, void* dpre_nhwc32, int N, int H, int W, void* stream);
int tg_up2_bilinear_bwd(int dtype, const void* ddst, const void* lrelu_mask, void* dsrc, int N, int H, int W, int C, void* stream);

/* ---- opt-in VGG feature loss (code/train.py:30-45,124-127,253-273; code/ops.py:144-213) ----------------- */
/* The reference's VGG path cannot execute; DESIGN.md fixes its semantics.  The VGG-19 convolutions run
 * on tg_conv / tg_conv3x3_rw; these are the HBM-bound pieces between them.
 * tg_vgg_input: dst[n][y][x][c] = scale * src[n][c][y][x] + shift3[c] for c < 3, 0 for 3 <= c < 32
 *   (deprocess(x) * 255 - VGG_MEAN with scale = 127.5, shift3[c] = 127.5 - VGG_MEAN[c], code/train.py:31-32). */
int tg_vgg_input(int dtype, const float* src_nchw, void* dst_nhwc32, int N, int H, int W, float scale, const float* shift3,
                 void* stream);
/* per pixel p of [npix][C] (C % 32 == 0): cos_p = <g,t> / (sqrt(sum g^2 + 1e-12) sqrt(sum t^2 + 1e-12));
 * *acc += sum_p cos_p;  dg[p][c] = coef * d cos_p / d g[p][c], zeroed where g <= 0 when relu_mask (g is a ReLU output).
 * (code/train.py:258-266 with the per-pixel channel norm of code/train.py:39-40 as DESIGN.md fixes it). */
int tg_cosine_loss(int dtype, const void* fg, const void* ft, void* dg, int64_t npix, int C, float coef, int relu_mask,
                   float* acc, const float* loss_scale, void* stream);
/* backward of nn.MaxPool2d((2,2), stride=2) (code/ops.py:149): out[n][y][x][c] = (res ? res : 0) + (dpool of the window
 * if (y,x) is the window's first maximum of `a`), zeroed where a <= 0 when relu_mask == 1, times 0.2 there when relu_mask == 2
 * (LeakyReLU(0.2): f_net's encoder blocks, code/models.py:6-11).  a, res, out [N][H][W][C],
 * dpool [N][H/2][W/2][C]. */
int tg_maxpool2_bwd(int dtype, const void* a, const void* dpool, const void* res, void* out, int N, int H, int W, int C,
                    int relu_mask, void* stream);
/* dpre[n][y][x][c] += dx[n][y][x][c] * scale * g (1 - g), g = gen[n][c][y][x], c < 3: chains d(loss)/d(vgg input) through
 * tg_vgg_input and the generator's output sigmoid (code/models.py:86) into d(loss)/d(pre-sigmoid); bias_acc3[c] (nullable)
 * += the per-channel sum of what was added (the output layer's bias gradient, as in tg_content_loss). */
int tg_vgg_input_grad(int dtype, const void* dx_nhwc32, const float* gen_nchw, void* dpre_nhwc32, int N, int H, int W,
                      float scale, float* bias_acc3, void* stream);

/* ---- data ingest: GPU-side frame resize (code/dataloader.py:84-88 of the reference: torchvision resize of PIL frames) --- */
/* PIL Image.resize(BILINEAR) + ToTensor, bit-exact: frames uint8 [N][H][W][3] -> tmp uint8 [N][H][OW][3] (horizontal pass)
 * -> out fp32 [N][3][OH][OW] = value / 255 (vertical pass).  bounds_* int32 [out][2] (first input index, count), kk_* int32
 * [out][ksize] 22-bit fixed-point coefficients, built on the host exactly as Pillow's Resample.c builds them
 * (pytorch-tecogan_amd/resize.py). */
int tg_resample_u8(const void* frames_u8, void* tmp_u8, float* out_nchw, const int32_t* bounds_w, const int32_t* kk_w,
                   int ksize_w, const int32_t* bounds_h, const int32_t* kk_h, int ksize_h, int N, int H, int W, int OH,
                   int OW, void* stream);

/* ---- step schedule support (no reference counterpart: the reference runs everything on one CUDA stream) ------- */
/* Creates a stream confined to every CU except the first `reserve_cus` CU-mask bits (hipExtStreamCreateWithCUMask; 64 bits
 * = 8 CUs on each of the 8 XCDs of an MI355X).  The dense lane of the step runs there, so that the generator chain's
 * 64-workgroup launches on an ordinary stream always find idle CUs.  *stream_out is a hipStream_t. */
int tg_stream_create_cumask(int reserve_cus, void** stream_out);
int tg_stream_destroy(void* stream);

#ifdef __cplusplus
}
#endif
#endif
