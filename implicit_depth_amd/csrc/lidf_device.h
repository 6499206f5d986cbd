// lidf_device.h — shared device-side definitions for liblidf_hip (gfx950 only).
//
// Weight-stream format
// --------------------
// The decoders run as a chain of v_mfma_f32_32x32x2_f32 instructions computed TRANSPOSED:
//   D[out feature][point] += A[out feature][k] * B[k][point]
// so that the accumulator registers of layer l are, without any data movement, the B operands of
// layer l+1 (lane l holds column `l&31` = one point; register r of a 32x32 tile holds feature row
// (r&3) + 8*(r>>2) + 4*(l>>5)).  The k order of every layer is therefore dictated by the register
// layout, and the nn.Linear weights are re-packed on the device, every call, into a "stream" of A
// fragments in exactly the order the kernel consumes them:
//   fragment = 64 floats, lane l holds W[32*tile + (l&31)][in(kstep, l>>5)]
//   quad     = 4 consecutive k-steps of one output tile, stored lane-major (float4 per lane, 1 KiB)
// Biases ride along as one extra k-step whose B operand is 1.0 in lanes 0..31 and 0 in 32..63.
//
// One block per decoder ("net"), consumed strictly front to back by an 8-deep register ring:
//   [ layer 1 : NL1 quads ][ u : 2 quads ][ layer 2 : 33 k-quads x 4 tiles ][ layer 3 : 17 x 2 ]
//   (layers 2 and 3 are k-quad major: every output tile advances together, so an input tile of 16
//    registers is produced right before its four k-quads and is dead after them)
//   layer 1, fused mode : for octave pair it, tile t, jq<3 : quad (it*24 + t*3 + jq) = k-steps
//                         4jq..4jq+3 of the 12 (sin x,y,z, cos x,y,z of octaves 2it, 2it+1);
//                         then 8 tail quads (x, y, z, pad), one per tile
//   layer 1, rows modes : for kq, tile t : quad (kq*nt + t); k-step jj of quad kq multiplies operand
//                         column 8kq + 4*half + jj, so the two half-waves read adjacent 16-byte
//                         pieces of the same row; column D carries the bias (operand 1.0)
// Every section is a multiple of 8 quads so that the ring index is static at every code point.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// "Last workgroup reduces" without a release fence (lidf_aux.hip: depth metrics, the two fingerprint kernels;
// lidf_frame.hip / lidf_device.h: the decoupled look-back): a workgroup publishes its partial result with an
// agent-scope atomic read-modify-write, keeps the RETURNED value alive (asm volatile) and only then takes its
// ticket with a second agent-scope atomic; the last arriver reads the partials with agent-scope atomics. The
// HIP memory model does not promise that order for relaxed atomics — the hardware this library is built for
// does: on gfx942 / gfx950 an atomic with return is performed at the device-coherent level (L2 / memory side
// for the other XCDs) before its value comes back, and every access involved is such an atomic, so nothing is
// read from a non-coherent cache. __ATOMIC_RELEASE on the ticket would be the portable form; on this device
// it is a write-back of the whole L2 (megabytes of the frame's dirty outputs: 24 -> 11 us for the metrics
// kernel, measured). Another target must not inherit the shortcut silently:
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "liblidf_hip orders its ticket reductions by returned atomic values (gfx942/gfx950 behaviour); use release/acquire on other targets"
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));  // dword-aligned 16-byte load

#define LIDF_H1 256   // gf_dim*4
#define LIDF_H2 128   // gf_dim*2
#define LIDF_H3 64    // gf_dim
// Activations the training forward keeps, per pass and n rows, as plane offsets in floats per row:
// H1 [n,256] | H2 [n,128] | H3 [n,64] | offset in [n] | sign words of H1 [n,8] | of H2 [n,4]
#define LIDF_ACT_OIN (LIDF_H1 + LIDF_H2 + LIDF_H3)
#define LIDF_ACT_M1 (LIDF_ACT_OIN + 1)
#define LIDF_ACT_M2 (LIDF_ACT_M1 + 8)
#define LIDF_ACT_ROW_FLOATS (LIDF_ACT_M2 + 4)
// partial block of the weight-gradient reduction (lidf_wgrad2_kernel / lidf_wgrad_reduce_kernel): a 128 x 256 block of
// C and two runs of 256 sums taken through the vector unit: the bias sums (column sums of A: 128; swapped launch:
// column sums of B: 256) and the products with ONE extra column of the other operand
#define LIDF_WG_SLAB (128 * 256 + 512)
#define LIDF_RING 8
#define LIDF_L2_QUADS 33   // per output tile: (128 k-steps + 1 bias k-step) / 4, rounded up
#define LIDF_L3_QUADS 17   // per output tile: (64 k-steps + 1 bias k-step) / 4, rounded up
#define LIDF_U_QUADS 2     // 8 fragments: u vector of the IEF rank-1 term, one per layer-1 tile
#define LIDF_PASS_QUADS (LIDF_U_QUADS + 4 * LIDF_L2_QUADS + 2 * LIDF_L3_QUADS)  // 168
#define LIDF_AUX_FLOATS 72  // per net: w4 by (half, reg) [2][32], b4 at [64]
#define LIDF_H_NK1 6          // split-f16 kernel: sin/cos k-steps of 16 per layer-1 tile (<= 8 octaves)
#define LIDF_HPASS_QUADS 288  // split-f16 decoder section: 278 quads padded to 18 chunks of 16
#define LIDF_MAX_L_FUSED 16 // octaves of the in-kernel positional encoding

enum { LIDF_MODE_FUSED = 0, LIDF_MODE_ROWS = 1, LIDF_MODE_L1ONLY = 2, LIDF_MODE_LINEAR = 3,
       LIDF_MODE_FUSED_H = 4,   // split-f16 stream of lidf_points_h.hip
       LIDF_MODE_TRAIN = 5,     // rows mode (stream of LIDF_MODE_ROWS) that keeps the activations
       LIDF_MODE_ROWS_GATHER = 6,    // rows mode whose layer-1 accumulators start from gathered rows
       LIDF_MODE_PNET_CHAIN = 7,     // pack jobs only: the per-point chain stream of a PointNet2Stage (PN_* below)
       LIDF_MODE_IEF16 = 8,          // pack jobs only: the stage-2 decoder's 16 x 16 x 4 stream (lidf_ief16.hip)
       LIDF_MODE_PNET_BWD = 9,       // pack jobs only: the backward chains' stream of a PointNet2Stage (PNB_* below)
       LIDF_MODE_CHAIN16 = 10 };     // the any-width decoder chain's stream (lidf_chain16.hip; L1Map.nt = gf / 16)
#define IEF16_PASS_QUADS 168   // 2 + 1 bias quads, 4 u quads, 128 layer-2 quads, 32 layer-3 quads, 1 padding quad
#define IEF16_AUX_FLOATS 72    // w4 [64] | b4 [1] (+ padding)

// One decoder's parameters as the packer sees them.
struct NetW {
    const float *w1, *b1, *w2, *b2, *w3, *b3, *w4, *b4, *wenc, *benc;
    int ld1;     // row stride of w1 (= d_in)
    int is_ief;  // w1 has 16 extra columns at [dcore, dcore+16)
    int dcore;   // D: number of "real" input columns
};

// How the columns of the layer-1 operand map onto w1 columns.
struct L1Map {
    // rows modes: operand column x in [0,n0) -> w1 column c0+x ; x in [n0,n0+n1) -> c1+(x-n0)
    int n0, c0, n1, c1;
    int D;         // n0+n1
    int KQ1;       // quads per tile = ceil((D+1)/8)
    int add_bias;  // bias k-step carries b1 (+ IEF constant) ; else 0
    int nt;        // output tiles per k-quad in the rows modes (8 for the decoders' layer 1)
    int nout;      // rows of w1 (outputs); tiles beyond it are zero
    int transposed;  // rows modes: weight element (out, col) is read at w1[col*ld1 + out] (dgrad); 2: operand
                     // columns >= n0 read the SECOND net of the pack call (its own w1 / ld1) at row c1 + (x - n0)
    int add_u;       // rows modes: operand column D+1 carries the IEF vector u (operand = offset)
    int xcol;        // linear mode: the last of the nt quads per k-quad carries ONE output (index 32 (nt - 1)), the
                     // same four weights for every row lane of a half-wave (lidf_linear_kernel<.., XCOL>)
    // fused mode
    int L;         // octaves
    int enter_c0;  // w1 column of enter-position embedding
    int leave_c0;  // w1 column of leave-position embedding
};

struct StreamLayout {
    int nets;       // 1 or 2
    int mode;       // LIDF_MODE_*
    int l1_quads;   // per net
    int net_quads;  // per-net block size in quads (layer 1 only for L1ONLY)
    int total;      // floats
    // guarded packing (lidf_*_pack_guarded_f32): when set, the pack kernel returns at once unless
    // guard->dirty != 0 — the parameters' fingerprint did not change since the stream was built
    const struct LidfPackGuardState* guard;
};

#define LIDF_FP_MAX_SEGS 24   // parameter buffers per fingerprint launch (two decoders: 18)
#define LIDF_FP_MULTI_SEGS 56 // ... of the frame's one fingerprint launch over all its modules (18 + 12 + 12 + 10)
#define LIDF_FP_GROUPS 4      // ... which keeps one guard per module group

// Device-side state of a guarded pack (lidf_pack_guard_bytes() bytes, zero-filled by the caller once):
// the fingerprint the packed streams were built from and the verdict of the latest comparison.
struct LidfPackGuardState {
    unsigned long long hash;   // fingerprint of the parameters the streams were packed from
    unsigned long long acc;    // running sum of the fingerprint launch in flight (0 between launches)
    unsigned int ticket;       // blocks of that launch that have finished
    int dirty;                 // verdict of the latest launch: 1 = re-pack
    int valid;                 // hash holds a fingerprint
    int pad;
};

// One stream to pack (lidf_pack_multi_kernel packs up to LIDF_PACK_JOBS of them per launch).
#define LIDF_PACK_JOBS 13   // (13 x 304 B of kernel arguments: under the 4 KiB limit; a frame with stage 2 has 22: two launches)
struct PackJob {
    StreamLayout lay;
    NetW n0, n1;
    L1Map m;
    float* stream;
    float* aux;
};
struct PackJobs {
    PackJob job[LIDF_PACK_JOBS];
    int n;
};


// ---- stream of the per-point chains of PointNet2Stage (lidf_pointnet.hip), 1 KiB quads consumed in order:
//   P1  1 quad            K = 6 inputs + bias (operand columns 4h + {0..3}: 6 = 1.0, 7 = 0)
//   P2  2 x 5 quads       K = 32 + bias, quad = 2 kq + t
//   P3  4 x 8 quads       K = 64 (second half of W3's columns), quad = 8 T + kq   [stage 2]
//   P4  4 x 17 quads      K = 128 + bias, quad = 17 T + kq                         [stage 2]
//   (+ 1 padding quad)
#define PN_P1 1
#define PN_P2 10
#define PN_P3 32
#define PN_P4 68
// per tile a stage walks a multiple of the ring's 8 quads, so that the ring slot of quad 0 is the
// same for every tile (stage 1 skips over five quads of P3, stage 2 over one padding quad)
#define PN_S1_QUADS 16
#define PN_S2_QUADS 112
struct PnetW {
    const float *w_p1, *b_p1, *w_p2, *b_p2, *w_p3, *w_p4, *b_p4;
};
// ---- stream of the backward chains (lidf_pointnet_train.hip): transposed weights in the same fragment order
//   A  64 quads  W4^T: dz4[o] += W4[k][o] dz5[k], quad = 16 T + kq (T: tile of o, kq over the 128 features k)
//      32 quads  W3[:, 64:]^T: df2[o] += W3[k][64 + o] dz4[k], quad = 16 t + kq
//   B   8 quads  W2^T: dz1[o] += W2[k][o] dz2[k] (kq over the 64 features k)
//       4 quads  W1^T padded to 32 rows: d inp[o] += W1[k][o] dz1[k], o < 6
//       4 padding quads
#define PNB_A_QUADS 96
#define PNB_B_QUADS 16
#ifdef __HIPCC__
__device__ __forceinline__ int pn_feature(int s, int half) {
    const int T = s >> 4, r = s & 15;
    return 32 * T + (r & 3) + 8 * (r >> 2) + 4 * half;
}
// element e of the chain stream (PN_S2_QUADS * 256 floats)
__device__ __forceinline__ float pn_stream_value(const PnetW& w, int e) {
    int quad = e / 256;
    const int lane = (e % 256) / 4, jj = e & 3;
    const int half = lane >> 5, c32 = lane & 31;
    float v = 0.f;
    if (quad < PN_P1) {
        const int x = 4 * half + jj;                       // operand column
        if (x < 6) v = w.w_p1[c32 * 6 + x];
        else if (x == 6) v = w.b_p1[c32];
    } else if (quad < PN_P1 + PN_P2) {
        quad -= PN_P1;
        const int kq = quad / 2, t = quad % 2, s = 4 * kq + jj, out = 32 * t + c32;
        if (s < 16) v = w.w_p2[out * 32 + pn_feature(s, half)];
        else if (s == 16 && half == 0) v = w.b_p2[out];
    } else if (quad < PN_P1 + PN_P2 + PN_P3) {
        quad -= PN_P1 + PN_P2;
        const int T = quad / 8, kq = quad % 8, s = 4 * kq + jj, out = 32 * T + c32;
        v = w.w_p3[out * 128 + 64 + pn_feature(s, half)];  // columns 64..127 multiply f2
    } else if (quad < PN_P1 + PN_P2 + PN_P3 + PN_P4) {
        quad -= PN_P1 + PN_P2 + PN_P3;
        const int T = quad / 17, kq = quad % 17, s = 4 * kq + jj, out = 32 * T + c32;
        if (s < 64) v = w.w_p4[out * 128 + pn_feature(s, half)];
        else if (s == 64 && half == 0) v = w.b_p4[out];
    }
    return v;
}
// element e of the backward chains' stream ((PNB_A_QUADS + PNB_B_QUADS) * 256 floats)
__device__ __forceinline__ float pn_bwd_stream_value(const PnetW& w, int e) {
    const int quad = e / 256;
    const int lane = (e % 256) / 4, jj = e & 3;
    const int half = lane >> 5, c32 = lane & 31;
    float v = 0.f;
    if (quad < 64) {
        const int T = quad / 16, kq = quad % 16;
        v = w.w_p4[pn_feature(4 * kq + jj, half) * 128 + 32 * T + c32];
    } else if (quad < 96) {
        const int q = quad - 64, t = q / 16, kq = q % 16;
        v = w.w_p3[pn_feature(4 * kq + jj, half) * 128 + 64 + 32 * t + c32];
    } else if (quad < 104) {
        const int kq = quad - 96;
        v = w.w_p2[pn_feature(4 * kq + jj, half) * 32 + c32];
    } else if (quad < 108) {
        const int kq = quad - 104;
        if (c32 < 6) v = w.w_p1[pn_feature(4 * kq + jj, half) * 6 + c32];
    }
    return v;
}
#endif  // __HIPCC__

static inline int lidf_l1_quads(int mode, const L1Map& m) {
    if (mode == LIDF_MODE_FUSED) return 24 * ((m.L + 1) / 2) + 8;
    return m.KQ1 * m.nt;
}

static inline StreamLayout lidf_make_layout(int nets, int mode, const L1Map& m) {
    StreamLayout s;
    s.nets = nets;
    s.mode = mode;
    s.l1_quads = lidf_l1_quads(mode, m);
    s.net_quads = s.l1_quads + (mode == LIDF_MODE_L1ONLY || mode == LIDF_MODE_LINEAR ? 0 : LIDF_PASS_QUADS);
    s.total = nets * s.net_quads * 256;
    s.guard = nullptr;
    return s;
}

static inline StreamLayout lidf_make_layout_h(int nets, const L1Map& m) {
    StreamLayout s;
    s.nets = nets;
    s.mode = LIDF_MODE_FUSED_H;
    (void)m;
    s.l1_quads = 0;  // layer 1 is interleaved with layer 2 inside the section
    s.net_quads = LIDF_HPASS_QUADS;
    s.total = nets * s.net_quads * 256;
    s.guard = nullptr;
    return s;
}

// Arguments of the per-point kernel (all three modes).
struct PointsArgs {
    const float* stream;
    const float* aux;
    int nets;           // per-net blocks in the stream
    int l1_quads, net_quads;
    long long n;        // points (fused) or rows
    const int* n_dev;   // optional device-side count overriding n (then n is the capacity the launch is
                        // sized for): the sync-free frame path, where sizes never reach the host
    // per net: passes (n_iter for IEF, 1 for IMNet), initial value (0.001 IEF / 0 IMNet),
    // output activation, output pointer ([n] or NULL), whether it is the offset net
    int npass[2];
    float init[2];
    int sigmoid[2];
    float* out[2];
    int is_offset[2];   // fused: net whose output drives pair_pred_pos
    // rows modes
    const float* X;
    long long ldx;
    int D, KQ1, has_bias;
    float* out_base;    // L1ONLY: [n, nets*256]
    // fused mode
    const int* pair_ray;
    const int* pair_vox;
    const float* pair_t;
    const float* ray_dir;
    const float* voxpart;   // [V, nets*256]
    const float* raypart;   // [R, nets*256]
    int part_ld, part_off;  // fused mode: row stride / first column (floats) of this launch's nets inside the two
                            // tables (0, 0 = nets * 256, 0; one net of a two-net table: 512 and 0 or 256)
    const float* vox_center;
    int pos_rel, L;
    float r0, rscale, sqrt3, part_size;
    float* pair_pred_pos;   // [n,3]
    int* tile_counter;      // split-f16 kernel: dynamic tile hand-out (zeroed by its packer)
    int dyn_min_tiles;      // fused f32 kernel: dynamic hand-out from this many wave-tiles per wavefront (0: 32)
    int dyn_chunk;          // ... in chunks of this many wave-tiles (0: LIDF_CHUNK)
    int tail_split;         // fused f32 kernel, static split, two nets: the tiles of the partial last round are
                            // handed out net by net (see points_fused_body)
    // LIDF_MODE_TRAIN (one net): X = the per-pair layer-1 operand rows, voxpart[pair_vox] and
    // raypart[pair_ray] are added to layer 1; pass k keeps H1 | H2 | H3 | offset-in | sign words at
    // tr_passes + k * tr_pass_floats (the LIDF_ACT_* planes above), the pre-activation output
    // goes to tr_pre [n]
    // (per net; the fused training forward keeps both nets' activations in one launch)
    float* tr_passes[2];
    long long tr_pass_floats;
    float* tr_pre[2];
};

// Arguments of the generic linear-layer kernel (lidf_linear.hip): out = epilogue(X W^T + b).
struct LinearArgs {
    const float* stream;   // rows-mode layer-1 section with `nt` tiles: quad (kq*nt + t)
    int kq1;
    const float* X;        // [n, D] rows, row stride ldx
    long long ldx, n;
    const int* n_dev;      // optional device-side row count overriding n (n = capacity of the launch)
    int nt_total;          // set by the launcher: > 0 = launch split over the output tiles (grid.y)
    int D, has_bias;
    const float* addrows;  // optional: += addrows[addidx[row], 0:32*nt]
    const int* addidx;
    int ld_add;
    int relu;
    float* out;            // optional store, row stride ld_out
    long long ld_out;
    float* pool;           // optional: pool[poolidx[row], f] = max(pool[..], value); needs relu
    const int* poolidx;
    int ld_pool;
    // training path (lidf_train.hip)
    float slope;           // with relu: leaky slope (0 = plain ReLU)
    const float* xoff;     // optional [n]: operand of column D+1 (the IEF offset fed to this pass)
    const float* mask_src; // optional [n, ld_mask]: value *= (mask_src > 0 ? 1 : mask_slope)
    long long ld_mask;
    float mask_slope;
    int nout;              // > 0: only columns < nout are stored
    int accumulate;        // out += value instead of out = value
    const float* addrows2; // optional second gathered term: += addrows2[addidx2[row]]
    const int* addidx2;
    int ld_add2;
    // operand rows from two buffers (the decoder pair's joint input gradient): k-quads [0, kq_split) read X, the
    // k-quads behind them X2 (row stride ldx2, its column 0 = operand column 8 kq_split); D % 8 == 0
    int kq_split;
    const float* X2;
    long long ldx2;
    // one more output column (index 32 nt) behind the nt tiles, through the vector unit: the stream carries nt + 1
    // quads per k-quad (L1Map.xcol); plain or accumulating store only, nout = 32 nt
    int xcol;
};

// Regular voxel grid of LIDF.get_occ_vox_bound (models/pipeline.py:162-201): lower corner (already
// widened by half a voxel), voxel size, cells per axis, frames.
struct GridSpec {
    float xmin[3];
    float crop;
    int r[3];
    int B;
};

// Optional cell table of the stepwise end-voxel lookup (lidf_refine.hip: lidf_refine_endvox_cells_kernel).
struct CellLookup {
    GridSpec g;
    const int* coord;   // [V,3] cell of every voxel
    int* table;         // [B * r0 * r1 * r2] cell -> largest voxel index, -1 = empty
    int ready;          // the table already holds this voxel list
};

// Arguments of one refine iteration's per-ray launch (lidf_refine.hip: lidf_refine_step_kernel).
struct RefineStepArgs {
    const float* prev_pos;     // [R,3]
    const float* prev_off;     // [R] offsets of the previous iteration, or NULL (first iteration)
    float r0, rs;
    float* cur_pos;            // [R,3] written when prev_off (else the iteration starts from prev_pos)
    const float* ray_dir;
    const int *ray_bid, *ray_flat;
    const long long* max_pair_id;
    const int* pair_vox;
    const float* vbound;
    const int* vox_bid;
    GridSpec g;
    const int *cell_flag, *cell_rank;
    const float* rgb;
    long long hw;
    int pnet_rel, pos_rel, L;
    float* pnet_inp;           // rows behind the *row0_dev valid points
    int* pnet_vox;
    const unsigned char* sel;
    int* end_voxel;
    float* inp_embed;          // embed(pos) lands at columns [256, 256 + E)
    int ld_e;
    const int* dims;           // device {R, P, V}
    const int* row0_dev;
    float *zero0, *zero1;
    long long nzero0, nzero1;
};

// Arguments of the stage-2 decoder on 16 x 16 x 4 tiles (lidf_ief16.hip).
struct Ief16Args {
    const float* stream;   // (16 KQ + IEF16_PASS_QUADS) KiB
    const float* aux;      // IEF16_AUX_FLOATS
    int KQ;                // layer-1 k-quads of 16 columns
    int E;                 // embed(pos) columns (operand columns beyond it are zero)
    long long n;           // rays
    const int* n_dev;      // optional device-side count (n = capacity)
    const float* X;        // [n, ldx]: embed(pos) columns start at X (the caller offsets the row)
    long long ldx;
    const int* vox;        // [n] row of voxpart (the end voxel)
    const float* voxpart;  // [V, 256]  W1[:, vox feat] f + b1 (+ c)
    const float* raypart;  // [n, 256]  W1[:, ROI | dir] rayfeat
    int npass;
    float init;
    int sigmoid;
    float* out;            // [n]
};

// Arguments of the any-width decoder chain (lidf_chain16.hip); G = gf / 16, table rows are 4 gf = 64 G floats.
struct Chain16Args {
    const float* stream;   // (KQ * 4 G + lidf_chain16_pass_quads(G)) KiB
    const float* aux;      // w4 [16 G] | b4 [1] (lidf_chain16_aux_floats(G))
    int KQ;                // layer-1 k-quads of 16 columns
    int E;                 // per-row operand columns (columns beyond it are zero)
    long long n;           // rows
    const float* X;        // [n, ldx]; readable up to column 16 KQ of every row
    long long ldx;
    const int* vox;        // [n] row of voxpart, or NULL (row 0 for every row)
    const float* voxpart;  // [V, 64 G]  W1[:, per-voxel columns] f + b1 (+ c), or NULL
    const int* ray;        // [n] row of raypart, or NULL (row r)
    const float* raypart;  // [R, 64 G], or NULL
    int npass;
    float init;
    int sigmoid;
    float* out;            // [n]
};
static inline int lidf_chain16_pass_quads(int G) {
    const int T1 = 4 * G, T2 = 2 * G, T3 = G;
    const int raw = (T2 + 3) / 4 + T1 * T2 + T1 / 4 + (T3 + 3) / 4 + T2 * T3;
    return (raw + 7) / 8 * 8;
}
static inline int lidf_chain16_aux_floats(int G) { return 16 * G + 8; }

// Arguments of the two-layer per-voxel kernel (lidf_linear.hip: lidf_vox2_kernel).
struct Vox2Args {
    const float* X;          // [n, D1] rows, stride ldx
    long long ldx, n;
    const int* n_dev;        // optional device-side row count (n = capacity of the launch)
    const float* s1;         // layer 1: rows-mode stream, quad (kq * nt1 + t)
    int kq1, nt1, D1, bias1, relu1;
    float* out1;             // optional store [n, 32 nt1], stride ld1
    long long ld1;
    const float* s2;         // layer 2 (NULL: none): K = 32 nt1 (+ bias column), quad (kq * nt2 + t)
    int kq2, nt2, bias2, relu2;
    float* out2;             // [n, 32 nt2], stride ld2
    long long ld2;
};

#ifdef __HIPCC__
// sin/cos of x*2^o for the positional encoding (implicit_net.py:30-32: freq bands are exact powers
// of two). x/(2 pi) is formed once as hi + lo (fma residual + low part of 1/(2 pi)); scaling by
// 2^o and v_fract are exact, so the argument handed to v_sin_f32 / v_cos_f32 (which take
// revolutions) carries no octave-dependent error: |err| <= 4.2e-7 at every octave, measured
// against double precision (scripts/hwsin_test.hip). 5 VALU per (coordinate, octave).
struct Rev {
    float hi, lo;
};
__device__ __forceinline__ Rev to_rev(float x) {
    const float C_HI = 0.15915493667125702f;   // fl(1/(2 pi))
    const float C_LO = 6.4206383e-09f;         // 1/(2 pi) - C_HI
    Rev r;
    r.hi = x * C_HI;
    r.lo = fmaf(x, C_HI, -r.hi) + x * C_LO;
    return r;
}
__device__ __forceinline__ void rev_sincos(const Rev& r, float sc, float& s, float& c) {
    const float t = __builtin_amdgcn_fractf(r.hi * sc) + r.lo * sc;
    s = __builtin_amdgcn_sinf(t);
    c = __builtin_amdgcn_cosf(t);
}
// element k >= 3 of [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...] for one coordinate value q (the per-ray rows
// of stage 2 and the per-ray direction embedding: same arithmetic as the per-point kernel's embedding)
__device__ __forceinline__ float pe_value(float q, int k) {
    const int l = (k - 3) / 6;
    const float sc = (float)(1 << l);
    const Rev r = to_rev(q);
    const float t = __builtin_amdgcn_fractf(r.hi * sc) + r.lo * sc;
    return ((k - 3) % 6) < 3 ? __builtin_amdgcn_sinf(t) : __builtin_amdgcn_cosf(t);
}

#endif  // __HIPCC__


#ifdef __HIPCC__
__device__ __forceinline__ int block_scan_256(int v, int* s_tmp, int& total) {
    // inclusive scan of one value per thread over 256 threads; returns exclusive prefix
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const int o = __shfl_up(inc, s);
        if (lane >= s) inc += o;
    }
    if (lane == 63) s_tmp[wave] = inc;
    __syncthreads();
    int wpre = 0;
    for (int w = 0; w < wave; ++w) wpre += s_tmp[w];
    total = s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3];
    __syncthreads();
    return wpre + inc - v;
}

#endif  // __HIPCC__

#ifdef __HIPCC__
// ---- decoupled look-back: an exclusive prefix over the workgroups of ONE launch ------------------
// (the frame path: a count -> scan -> fill chain of three or five launches becomes one.) One status
// word per workgroup, zeroed before the launch: bits 63..62 = 0 nothing yet / 1 the workgroup's own
// aggregate / 2 the inclusive prefix up to and including it; the low 62 bits hold the value (one count,
// or two counts of < 2^31 packed side by side: fields add independently). A workgroup publishes its
// aggregate as soon as it has it, then one of its wavefronts walks back over its predecessors, 64 at a
// time, adding aggregates until it meets an inclusive prefix. Workgroups take their index from an
// atomic ticket (not blockIdx), so a predecessor is always a workgroup that already runs. The words are
// relaxed agent-scope atomics: a word carries its value itself, nothing else is published through it
// (release / acquire would write back and invalidate the L2 at every step).
#define LIDF_LB_AGG 1ull
#define LIDF_LB_INC 2ull
#define LIDF_LB_MASK 0x3fffffffffffffffull
__device__ __forceinline__ void lb_store(unsigned long long* st, long long i, unsigned long long flag,
                                         unsigned long long v) {
    __hip_atomic_store(st + i, (flag << 62) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long lb_wave_sum(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)v, o), hi = __shfl_xor((unsigned)(v >> 32), o);
        v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}
// exclusive prefix of workgroup `bid`; every lane of ONE wavefront calls it and receives the result
__device__ __forceinline__ unsigned long long lb_exclusive(const unsigned long long* st, long long bid,
                                                           int lane) {
    unsigned long long run = 0;
    for (long long j = bid - 1; j >= 0; j -= 64) {
        const long long idx = j - lane;
        unsigned long long w;
        for (;;) {
            w = idx >= 0 ? __hip_atomic_load(st + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                         : (LIDF_LB_INC << 62);
            if (__ballot((w >> 62) == 0) == 0) break;
            __builtin_amdgcn_s_sleep(1);
        }
        const unsigned long long inc = __ballot((w >> 62) == LIDF_LB_INC);
        const int first = inc ? __builtin_ctzll(inc) : 64;
        run += lb_wave_sum(lane <= first ? (w & LIDF_LB_MASK) : 0ull);
        if (inc) break;
    }
    return run;
}
// the workgroup's ticket (its index in the prefix order), broadcast through `s_bid`
__device__ __forceinline__ int lb_ticket(int* counter, int* s_bid) {
    if (threadIdx.x == 0) *s_bid = atomicAdd(counter, 1);
    __syncthreads();
    return *s_bid;
}
#endif  // __HIPCC__

#ifdef __HIPCC__
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per device and kernel: the attribute belongs
// to the device's copy of the function, the call costs tens of microseconds of host time, and the
// small launches of the evaluation path are host-bound. `done` is the call site's static table
// (a repeated set under a race is harmless).
static inline hipError_t lidf_max_lds_once(bool (&done)[64], const void* fn, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (done[dev]) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done[dev] = true;
    return e;
}
#endif  // __HIPCC__
