// gnnome_node_aggregate_stream_f32: both gated aggregations of SymGatedGCN and the node update with e' read ONCE.
//
//   s_p   = sigmoid(e[p,:])
//   fwd_i = sum_{p in in(i)}  s_p * A2h[src_p,:] / (sum_{p in in(i)}  s_p + 1e-6)
//   bwd_i = sum_{p in out(i)} s_p * A3h[dst_p,:] / (sum_{p in out(i)} s_p + 1e-6)
//   h'_i  = relu((A1h_i + fwd_i + bwd_i) * scale + shift) + h_i
//
// Reference lines replaced: gated_gcn_full.py:111-114 (in-edge sums on g), :124-127 (out-edge sums on dgl.reverse(g)),
// :129-137 (sum, bn_h in eval mode, relu, residual) - the same contract as gnnome_node_aggregate_f32 with the affine norm.
//
// Why a second kernel (round 5).  k_node_aggregate gives every node a wave that reads the e' rows of its
// in-edges (a contiguous run of destination-sorted positions) and then RE-READS the e' rows of its out-edges through out_pos:
// every row crosses the L1 twice, the second time as scattered 4H-byte pieces, and the launch sits at 3.5-3.8 TB/s for three
// rounds whatever is done inside it.  Here the e' rows are STREAMED in destination order, each exactly once, and a row's
// contribution to its SOURCE node (the out-edge half) is scattered into a window of accumulator slots in LDS instead of being
// gathered later:
//
//   * the node range is cut into chunks of consecutive destination nodes (balanced by rows); one workgroup walks one chunk;
//   * the H channels are cut into slices of CH; wave w of the workgroup owns slice w for the WHOLE chunk - the aggregation is
//     channel-wise, so waves never share an accumulator: no atomics, no barriers, no workgroup-level synchronisation at all;
//   * a wave walks the chunk's rows in STEPS of RPS = 64 / (CH / 4) rows that all belong to ONE destination node (distinct
//     sources, so the RPS lane groups of a step touch RPS different slots); a lane group keeps one row's 16-byte piece;
//   * in-edge half: per-lane-group partial sums in registers, combined across the groups with a fixed xor tree at the node's
//     last step;  out-edge half: slot[src] += (s * A3h[dst], s) in LDS;
//   * which slot a source owns, which row opens / closes it, and which event of a node is its last (that event runs the node
//     update) is a SCHEDULE computed once per graph (gnnome_build_stream_schedule): interval colouring of the nodes' live
//     ranges [first event, last event] with K slots, lowest free slot first - a function of the graph alone;
//   * rows whose source lies in another chunk, duplicates of a (src, dst) pair and rows of nodes that found no free slot are FAR:
//     the stream skips their out-edge half, the node becomes PENDING (its last event parks (A1h + fwd, num_b, den_b) in a side
//     buffer) and k_stream_finish adds the far rows in out-list order and runs the update.  On a layout-ordered assembly graph
//     4-13 % of the rows are far (chunk boundaries + the long-range "repeat" edges); the host falls back to k_node_aggregate
//     when the far fraction says the numbering has no locality.
// The summation order is a function of (graph, schedule parameters) only: results are bit-reproducible; they differ from
// k_node_aggregate's by fp32 reassociation.
#include "common.h"

namespace gnnome {

constexpr int kFar = 0xFF;             // edge_meta: the row's out-edge half is not done by the stream
constexpr int kStMiddle = 0, kStFirst = 1, kStLastFinal = 2, kStLastPending = 3;   // edge_meta >> 6
constexpr int kSlotUnalloc = 0xFF, kSlotOverflow = 0xFE;
constexpr int kLateDistance = 3;   // > the number of steps the stream requests a closing row's operands ahead of the row (2)
// step descriptor word 2
constexpr int kDLastStep = 1 << 5, kDFirst = 1 << 6, kDLast = 1 << 7, kDHasSlot = 1 << 8;

// chunk_node[c] = first node whose in-edge run starts at or after row c * E / C (chunk_node[0] = 0, chunk_node[C] = N)
__global__ void k_stream_chunk_bounds(const int32_t* __restrict__ in_ptr, int64_t n, int64_t e, int chunks, int32_t* __restrict__ chunk_node) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > chunks) return;
    if (c == 0) { chunk_node[0] = 0; return; }
    if (c == chunks) { chunk_node[c] = (int32_t)n; return; }
    const int64_t target = e * c / chunks;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (in_ptr[mid] < target) lo = mid + 1; else hi = mid;
    }
    chunk_node[c] = (int32_t)lo;
}

// Pass 1, one thread per node s: which of its out-edges are FAR (other chunk, or a repeat of the previous (s, d) pair), the last
// destination among the near ones, and a pending index if any row is far.
__global__ void k_stream_sched_nodes(int64_t n, int chunks, const int32_t* __restrict__ chunk_node, const int32_t* __restrict__ out_ptr,
                                     const int32_t* __restrict__ out_pos, const int32_t* __restrict__ out_dst, uint8_t* __restrict__ edge_meta,
                                     int32_t* __restrict__ last_near, uint8_t* __restrict__ slot_of, int32_t* __restrict__ node_pend,
                                     int32_t* __restrict__ pend_nodes, int32_t* __restrict__ counters) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    int lo = 0, hi = chunks;   // the chunk of s: the last c with chunk_node[c] <= s (empty chunks share a boundary: take the last)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (chunk_node[mid] <= s) lo = mid; else hi = mid;
    }
    const int n0 = chunk_node[lo], n1 = chunk_node[lo + 1];
    const int k0 = out_ptr[s], k1 = out_ptr[s + 1];
    int last = (int)s, far = 0;
    for (int k = k0; k < k1; ++k) {
        const int d = out_dst[k];
        const bool near = d >= n0 && d < n1 && !(k > k0 && out_dst[k - 1] == d);
        if (near) {
            last = max(last, d);
            edge_meta[out_pos[k]] = 0;
        } else {
            edge_meta[out_pos[k]] = kFar;
            ++far;
        }
    }
    last_near[s] = last;
    slot_of[s] = kSlotUnalloc;
    int idx = -1;
    // also pending: a node whose closing row follows its in-run end by fewer than kLateDistance steps - the stream requests a closing row's
    // operands that many steps ahead, and the parked A1h + fwd of a node that is NOT pending is read back from h_out
    if (far > 0 || (last > s && last - s < kLateDistance)) {
        idx = atomicAdd(&counters[0], 1);
        pend_nodes[idx] = (int32_t)s;
        if (far > 0) atomicAdd(&counters[1], far);
    }
    node_pend[s] = idx;
}

// Pass 2, one thread per chunk: the sweep over the chunk's destination nodes in order - slot allocation (lowest free slot first),
// the state of every near row, the step descriptors.
__global__ void k_stream_sched_chunks(int chunks, int rps, int num_slots, const int32_t* __restrict__ chunk_node, const int32_t* __restrict__ in_ptr,
                                      const int32_t* __restrict__ srt_src, const int32_t* __restrict__ last_near, uint8_t* __restrict__ slot_of,
                                      uint8_t* __restrict__ edge_meta, int32_t* __restrict__ node_pend, int32_t* __restrict__ pend_nodes,
                                      int32_t* __restrict__ counters, int4* __restrict__ steps, int32_t* __restrict__ chunk_steps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= chunks) return;
    const int n0 = chunk_node[c], n1 = chunk_node[c + 1];
    unsigned long long free_slots = num_slots >= 64 ? ~0ull : ((1ull << num_slots) - 1ull);
    int4* out = steps + ((int64_t)in_ptr[n0] / rps + n0);
    int nsteps = 0, overflowed = 0, max_live = 0;
    auto make_pending = [&](int s) {
        if (node_pend[s] < 0) {
            const int idx = atomicAdd(&counters[0], 1);
            pend_nodes[idx] = s;
            node_pend[s] = idx;
        }
    };
    for (int t = n0; t < n1; ++t) {
        const int ib = in_ptr[t], ie = in_ptr[t + 1];
        unsigned long long free_after = 0;
        for (int p = ib; p < ie; ++p) {
            if (edge_meta[p] == kFar) continue;
            const int s = srt_src[p];
            int so = slot_of[s];
            if (so == kSlotOverflow) {
                edge_meta[p] = kFar;
                atomicAdd(&counters[1], 1);
                continue;
            }
            int state = kStMiddle;
            if (so == kSlotUnalloc) {
                if (free_slots == 0) {
                    slot_of[s] = kSlotOverflow;
                    edge_meta[p] = kFar;
                    atomicAdd(&counters[1], 1);
                    make_pending(s);
                    ++overflowed;
                    continue;
                }
                so = __ffsll((long long)free_slots) - 1;
                free_slots &= ~(1ull << so);
                slot_of[s] = (uint8_t)so;
                state = kStFirst;
            }
            if (last_near[s] == t && s != t) {   // this row is the last event of its source
                state = node_pend[s] >= 0 ? kStLastPending : kStLastFinal;
                free_after |= 1ull << so;
            }
            edge_meta[p] = (uint8_t)((state << 6) | so);
        }
        // the node's own in-run end
        int so = slot_of[t], info = 0;
        const bool last_ev = last_near[t] == t;
        if (so == kSlotOverflow) {
            info = kDFirst | kDLast;
        } else if (so == kSlotUnalloc) {
            if (last_ev) {
                info = kDFirst | kDLast;
            } else if (free_slots == 0) {
                slot_of[t] = kSlotOverflow;   // its later near rows turn far
                make_pending(t);
                ++overflowed;
                info = kDFirst | kDLast;
            } else {
                so = __ffsll((long long)free_slots) - 1;
                free_slots &= ~(1ull << so);
                slot_of[t] = (uint8_t)so;
                info = kDFirst | kDHasSlot | (so << 16);
            }
        } else {
            info = kDHasSlot | (so << 16);
            if (last_ev) {
                info |= kDLast;
                free_after |= 1ull << so;
            }
        }
        max_live = max(max_live, num_slots - __popcll(free_slots));
        const int din = ie - ib, nst = din > 0 ? (din + rps - 1) / rps : 1;
        for (int j = 0; j < nst; ++j) {
            const int cnt = min(rps, din - j * rps);
            out[nsteps++] = make_int4(ib + j * rps, t, (cnt < 0 ? 0 : cnt) | (j == nst - 1 ? (kDLastStep | info) : 0), node_pend[t]);
        }
        free_slots |= free_after;
    }
    chunk_steps[c] = nsteps;
    if (overflowed) atomicAdd(&counters[2], overflowed);
    atomicMax(&counters[3], max_live);
    atomicMax(&counters[4], nsteps);
}

struct StreamArgs {
    const float* e;
    int H;
    const float *A1h, *A2h, *A3h;
    int ldn;
    const float* h_in;
    int ldh;
    float* h_out;
    const float *scale, *shift;
    const int32_t* in_ptr;
    const int32_t* srt_src;
    const uint8_t* edge_meta;
    const int4* steps;
    const int32_t* chunk_node;
    const int32_t* chunk_steps;
    const int32_t* node_pend;
    float* pend_rows;   // [num_pending][3][H]: A1h + fwd | num_b | den_b
    int chunks;
    int64_t num_nodes, num_edges;
};

constexpr int kStreamCH = 32;    // channels per wave
constexpr int kStreamRPS = 16;   // rows per step: 4 lanes x 8 channels per row
constexpr int kStreamSlots = 62;

// v + (v rotated right by R lanes inside its row of 16 lanes): one v_add_f32 with a DPP operand
template <int R>
__device__ __forceinline__ float add_row_ror(float v) {
    return v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x120 + R, 0xF, 0xF, false));
}
// "this register holds nothing worth keeping": ends the live range of a pipeline register without an instruction (a conditional
// assignment otherwise keeps the previous step's value alive around the whole loop, an unconditional zero costs a v_mov per register)
template <typename T>
__device__ __forceinline__ void forget(T& v) {
    asm volatile("" : "=v"(v));
}
__device__ __forceinline__ f32x4 load16(const float* base, unsigned byte_off) {   // uniform base + 32-bit lane offset: no 64-bit vector arithmetic
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + byte_off);
}

// One workgroup = one chunk, wave w = channels [32 w, 32 w + 32).  Lane (g = lane / 4, l = lane % 4) of a step holds row g's channels
// 4 l .. 4 l + 3 and 16 + 4 l .. 16 + 4 l + 3 of the slice (two 16-byte pieces; a load instruction covers 64 contiguous bytes of each row).
// Software pipeline over the chunk's steps, everything a step carries lives in ONE of five register sets (step j in set j % 5) and the
// loop is unrolled five times by hand, so that no value is copied while its load is in flight (a register copy waits for the load):
//   iteration q:  C   step q: sigmoid, in-edge partial sums, slot updates in LDS, closing rows, the node's in-run end
//                 X   descriptor and row pieces x of step q + 4 requested (the HBM stream: four steps = 8 KB per wave in flight)
//                 A0  source ids + schedule bytes of step q + 3 requested
//                 A   the gathers of step q + 2 requested (its source ids have arrived): A2h[src], A3h[node], the node-level operands
//                     A1h[node], h[node] (4 bytes per lane) and, for a closing row, h[src], the pending index or the parked A1h + fwd of
//                     its source (in h_out since that node's in-run end - at least kLateDistance steps ago, the schedule makes every
//                     node with a closer closing row pending, and a pending node parks in pend_rows)
// The vector-memory queue returns in order: stage C waits for step q's operands only, and the requests of steps q + 1 .. q + 3 stay in
// flight across it (the first version requested the closing rows' operands one step ahead, AFTER the stage-C stores they might depend on:
// every iteration then waited for the newest request and drained the queue - 1.9 us per step).
// Slots are zero whenever they are free (zeroed at the start, and again by the event that closes them), so a row never has to know
// whether it is the first to touch its slot.
// BIG: the node tables exceed 4 GB (64-bit gather addresses); otherwise a gather is a uniform base + a 32-bit lane offset.
template <int K, bool BIG>
__global__ __launch_bounds__(512) void k_aggregate_stream(
    const float* __restrict__ e, const int H, const float* __restrict__ A1h, const float* __restrict__ A2h, const float* __restrict__ A3h, const int ldn,
    const float* __restrict__ h_in, const int ldh, float* h_out, const float* __restrict__ scale, const float* __restrict__ shift,
    const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ srt_src, const uint8_t* __restrict__ edge_meta, const int4* __restrict__ steps,
    const int32_t* __restrict__ chunk_node, const int32_t* __restrict__ chunk_steps, const int32_t* __restrict__ node_pend, float* __restrict__ pend_rows,
    const int chunks, const int last_row) {
    constexpr int CH = kStreamCH, RPS = kStreamRPS;
    constexpr int SLOT = 2 * CH;               // floats per slot: num_b[CH] | den_b[CH]
    constexpr int WAVE_LDS = K * SLOT + 4 * 64;   // + the in-run end's transposition buffer [4 rows of lanes][64 values]
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 2, l = lane & 3;
    const int c0 = wave * CH;                 // the slice
    const int cl = c0 + 4 * l;                // this lane's first piece; the second one starts 16 channels on
    float* const W = lds + wave * WAVE_LDS;
    float* const S = W + 4 * l;                // + slot * SLOT (+ 16: second piece, + CH: den)
    float* const T = W + K * SLOT;
    const int chunk = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, chunks));
    const int n0 = chunk_node[chunk];
    const int ns = chunk_steps[chunk];
    if (ns <= 0) return;
    const int4* __restrict__ sd = steps + ((int64_t)in_ptr[n0] / RPS + n0);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int i = lane * 4; i < K * SLOT; i += 256) *reinterpret_cast<f32x4*>(W + i) = zero;
    // lane constants: byte offsets of this lane's first piece inside a step's rows / a node row
    const unsigned xoff = (unsigned)(g * H + cl) * 4u, noff = (unsigned)cl * 4u;
    // node-level lane roles (the in-run end): lane j < 32 owns channel c0 + j
    const int cj = c0 + (lane & 31);
    const float scj = scale[cj], shj = shift[cj];
    const f32x4 sc_l = *reinterpret_cast<const f32x4*>(scale + cl), sc_h = *reinterpret_cast<const f32x4*>(scale + cl + 16);
    const f32x4 sh_l = *reinterpret_cast<const f32x4*>(shift + cl), sh_h = *reinterpret_cast<const f32x4*>(shift + cl + 16);

    struct StepRegs {
        int4 d;                          // descriptor (wave-uniform)
        int src, meta;                   // A0
        f32x4 xl, xh, a2l, a2h, a3l, a3h;   // A
        f32x4 tsl, tsh, hil, hih;        //   (closing rows)
        int pi;
        float a1j, hnj;                  //   (node level)
    };
    StepRegs R0, R1, R2, R3, R4;
    const int4 empty = make_int4(0, n0, 0, -1);
    R0.d = R1.d = R2.d = R3.d = R4.d = empty;
    f32x4 nfl = zero, nfh = zero, dfl = zero, dfh = zero;

    // a gathered row piece of a node table: row * ld + this lane's channels
    auto gather = [&](const float* table, int row, int ld, f32x4& lo, f32x4& hi) {
        if (BIG) {
            const float* r = table + (int64_t)row * ld + cl;
            lo = *reinterpret_cast<const f32x4*>(r), hi = *reinterpret_cast<const f32x4*>(r + 16);
        } else {
            const unsigned off = (unsigned)(row * ld + cl) * 4u;
            lo = load16(table, off), hi = load16(table, off + 64u);
        }
    };
    // Every request of the pipeline is UNCONDITIONAL: hipcc's wait insertion counts only the requests it can prove were issued, so a
    // wait for an older load behind loads inside an `if (live)` is computed as if those did not exist and drains them (measured: the
    // same 1.9 us per step as with no pipeline at all).  A dead lane (g >= count) repeats the request of lane l of the step's first row,
    // which coalesces with that lane's; what it receives is never used.
    auto stage_x = [&](StepRegs& R, const int j, const int4 dj) {   // the descriptor and the row pieces of step j
        R.d = j < ns ? dj : empty;
        const int p0 = min(R.d.x, last_row);   // (an empty step of a trailing node without in-edges points one past the end)
        const float* xr = e + (int64_t)p0 * H;   // (wave-uniform)
        const unsigned off = g < (R.d.z & 31) ? xoff : noff;
        // the e rows are read once: nontemporal, so that they do not push the chunk's window of table rows (A2h[src]: every row gathered
        // ~10 times over ~25 steps) out of its share of the L2 - 64 chunks per XCD share 4 MB
        R.xl = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(xr) + off));
        R.xh = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(xr) + off + 64u));
    };
    auto stage_a0 = [&](StepRegs& R) {   // source ids and schedule bytes
        const int p0 = min(R.d.x, last_row);
        const int gg = g < (R.d.z & 31) ? g : 0;
        const int32_t* sp = srt_src + p0;
        const uint8_t* mp = edge_meta + p0;
        R.src = sp[gg];
        R.meta = mp[gg];
    };
    auto stage_a = [&](StepRegs& R) {
        gather(A2h, R.src, ldn, R.a2l, R.a2h);
        const float* br = A3h + (int64_t)R.d.y * ldn;
        R.a3l = load16(br, noff), R.a3h = load16(br, noff + 64u);
        R.a1j = A1h[(int64_t)R.d.y * ldn + cj];
        R.hnj = h_in[(int64_t)R.d.y * ldh + cj];
        // (the closing rows' operands stay conditional: a handful of requests in ~40 % of the steps)
        forget(R.tsl), forget(R.tsh), forget(R.hil), forget(R.hih), forget(R.pi);
        if (g < (R.d.z & 31) && R.meta != kFar && (R.meta >> 6) >= kStLastFinal) {
            if ((R.meta >> 6) == kStLastPending) {
                R.pi = node_pend[R.src];
            } else {
                gather(h_out, R.src, H, R.tsl, R.tsh);
                gather(h_in, R.src, ldh, R.hil, R.hih);
            }
        }
    };
    auto stage_c = [&](StepRegs& R) {
        if (g < (R.d.z & 31)) {   // (no vector-memory loads inside: see above)
            f32x4 sl, sh_;
#pragma unroll
            for (int k = 0; k < 4; ++k) sl[k] = sigmoidf_(R.xl[k]), sh_[k] = sigmoidf_(R.xh[k]);
            nfl += sl * R.a2l, nfh += sh_ * R.a2h;
            dfl += sl, dfh += sh_;
            if (R.meta != kFar) {
                float* slot = S + (R.meta & 63) * SLOT;
                f32x4 nbl = *reinterpret_cast<const f32x4*>(slot), nbh = *reinterpret_cast<const f32x4*>(slot + 16);
                f32x4 dbl = *reinterpret_cast<const f32x4*>(slot + CH), dbh = *reinterpret_cast<const f32x4*>(slot + CH + 16);
                nbl += sl * R.a3l, nbh += sh_ * R.a3h;
                dbl += sl, dbh += sh_;
                if ((R.meta >> 6) >= kStLastFinal) {   // the last event of the row's source: its node update, and the slot is free (= zero) again
                    if ((R.meta >> 6) == kStLastFinal) {
                        f32x4 yl, yh;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            yl[k] = relu_keep_nan((R.tsl[k] + nbl[k] * __builtin_amdgcn_rcpf(dbl[k] + kAggEps)) * sc_l[k] + sh_l[k]) + R.hil[k];
                            yh[k] = relu_keep_nan((R.tsh[k] + nbh[k] * __builtin_amdgcn_rcpf(dbh[k] + kAggEps)) * sc_h[k] + sh_h[k]) + R.hih[k];
                        }
                        float* o = h_out + (int64_t)R.src * H + cl;
                        *reinterpret_cast<f32x4*>(o) = yl;
                        *reinterpret_cast<f32x4*>(o + 16) = yh;
                    } else {   // a pending source: its in-run end has parked A1h + fwd in row 0 already
                        float* pr = pend_rows + ((int64_t)R.pi * 3 + 1) * H + cl;
                        *reinterpret_cast<f32x4*>(pr) = nbl, *reinterpret_cast<f32x4*>(pr + 16) = nbh;
                        *reinterpret_cast<f32x4*>(pr + H) = dbl, *reinterpret_cast<f32x4*>(pr + H + 16) = dbh;
                    }
                    nbl = nbh = dbl = dbh = zero;
                }
                *reinterpret_cast<f32x4*>(slot) = nbl, *reinterpret_cast<f32x4*>(slot + 16) = nbh;
                *reinterpret_cast<f32x4*>(slot + CH) = dbl, *reinterpret_cast<f32x4*>(slot + CH + 16) = dbh;
            }
        }
        if (R.d.z & kDLastStep) {   // (wave-uniform) the node's in-run ends with this step
            // sums over the 16 lane groups: two DPP rotations leave every lane with the sum of its row of 16 lanes' four groups, the four
            // row sums meet in LDS, TRANSPOSED: afterwards lane j < 32 holds sum s*A2h and lane 32 + j sum s of channel c0 + j, so that the
            // node update below runs on one element per lane instead of on 4 lanes of 64
            f32x4 v[4] = {nfl, nfh, dfl, dfh};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) v[i][k] = add_row_ror<8>(add_row_ror<4>(v[i][k]));
            if ((g & 3) == 0) {
                float* tr = T + (g >> 2) * 64 + 4 * l;
                *reinterpret_cast<f32x4*>(tr) = v[0], *reinterpret_cast<f32x4*>(tr + 16) = v[1];
                *reinterpret_cast<f32x4*>(tr + 32) = v[2], *reinterpret_cast<f32x4*>(tr + 48) = v[3];
            }
            nfl = nfh = dfl = dfh = zero;
            const float tot = (T[lane] + T[64 + lane]) + (T[128 + lane] + T[192 + lane]);
            const float den = __shfl_xor(tot, 32);   // lanes < 32: (tot, den) = (sum s*A2h, sum s)
            const float t = R.a1j + tot * __builtin_amdgcn_rcpf(den + kAggEps);
            const int node = R.d.y;
            if (lane < 32) {
                float* slot = W + ((R.d.z >> 16) & 63) * SLOT + lane;
                if (R.d.w >= 0) {   // a pending node: A1h + fwd goes to its side rows now; num_b / den_b follow with its last event
                    float* pr = pend_rows + (int64_t)R.d.w * 3 * H + cj;
                    pr[0] = t;
                    if (R.d.z & kDLast) {
                        float nb = 0.f, db = 0.f;
                        if (R.d.z & kDHasSlot) {
                            nb = slot[0], db = slot[CH];
                            slot[0] = 0.f, slot[CH] = 0.f;
                        }
                        pr[H] = nb, pr[2 * H] = db;
                    }
                } else if (R.d.z & kDLast) {
                    float nb = 0.f, db = 0.f;
                    if (R.d.z & kDHasSlot) {
                        nb = slot[0], db = slot[CH];
                        slot[0] = 0.f, slot[CH] = 0.f;
                    }
                    h_out[(int64_t)node * H + cj] = relu_keep_nan((t + nb * __builtin_amdgcn_rcpf(db + kAggEps)) * scj + shj) + R.hnj;
                } else {
                    h_out[(int64_t)node * H + cj] = t;   // parked until the node's closing row, at least kLateDistance steps from here
                }
            }
        }
    };

    stage_x(R0, 0, sd[0]), stage_x(R1, 1, sd[min(1, ns - 1)]), stage_x(R2, 2, sd[min(2, ns - 1)]), stage_x(R3, 3, sd[min(3, ns - 1)]);
    stage_a0(R0), stage_a0(R1), stage_a0(R2);
    stage_a(R0), stage_a(R1);
    int4 dnext = sd[min(4, ns - 1)];   // the descriptor of the step stage X takes next: a scalar load, one iteration ahead of its use
    // iteration q: C of step q; then X of q + 4, A0 of q + 3, A of q + 2
    auto iteration = [&](const int q, StepRegs& Rq, StepRegs& Rq2, StepRegs& Rq3, StepRegs& Rq4) {
        stage_c(Rq);
        stage_x(Rq4, q + 4, dnext);
        dnext = sd[min(q + 5, ns - 1)];
        stage_a0(Rq3);
        stage_a(Rq2);
    };
    for (int q = 0; q < ns; q += 5) {
        iteration(q, R0, R2, R3, R4);
        iteration(q + 1, R1, R3, R4, R0);
        iteration(q + 2, R2, R4, R0, R1);
        iteration(q + 3, R3, R0, R1, R2);
        iteration(q + 4, R4, R1, R2, R3);
    }
}

// The pending nodes: (A1h + fwd, num_b, den_b) parked by the stream + the far rows of the node's out-list in list order.
// One wave per pending node, H / 4 lanes per row, lane group g takes every G-th far row; fixed xor tree over the groups.
template <int H>
__global__ __launch_bounds__(256) void k_stream_finish(const float* __restrict__ e, const float* __restrict__ A3h, int ldn,
                                                       const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_pos,
                                                       const int32_t* __restrict__ out_dst, const uint8_t* __restrict__ edge_meta,
                                                       const int32_t* __restrict__ pend_nodes, const int32_t* __restrict__ counters,
                                                       const float* __restrict__ pend_rows, const float* __restrict__ h_in, int ldh,
                                                       float* __restrict__ h_out, const float* __restrict__ scale, const float* __restrict__ shift) {
    constexpr int LPR = H / 4, G = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int idx = blockIdx.x * 4 + wave;
    if (idx >= counters[0]) return;
    const int node = pend_nodes[idx];
    const int group = lane / LPR, c = (lane % LPR) * 4;
    const int k0 = out_ptr[node], k1 = out_ptr[node + 1];
    f32x4 nb = {0.f, 0.f, 0.f, 0.f}, db = nb;
    for (int base = k0; base < k1; base += 64) {
        // lane j looks at list entry base + j; the far ones are compacted in list order
        const int k = base + lane;
        int pos = 0, dd = 0;
        bool far = false;
        if (k < k1) {
            pos = out_pos[k];
            dd = out_dst[k];
            far = edge_meta[pos] == kFar;
        }
        unsigned long long m = __ballot(far);
        const int nfar = __popcll(m);
        for (int j0 = 0; j0 < nfar; j0 += G) {
            // the (j0 + group)-th set bit of m
            const int want = j0 + group;
            unsigned long long mm = m;
            for (int r = 0; r < want && mm; ++r) mm &= mm - 1;
            const bool live = want < nfar;
            const int srcl = live ? (__ffsll((long long)mm) - 1) : (__ffsll((long long)m) - 1);
            const int p = __shfl(pos, srcl), d = __shfl(dd, srcl);
            const f32x4 x = *reinterpret_cast<const f32x4*>(e + (int64_t)p * H + c);
            const f32x4 a3 = *reinterpret_cast<const f32x4*>(A3h + (int64_t)d * ldn + c);
            f32x4 s;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) s[kk] = live ? sigmoidf_(x[kk]) : 0.f;
            nb += s * a3;
            db += s;
        }
    }
    const float* pr = pend_rows + (int64_t)idx * 3 * H + c;
    const f32x4 t = *reinterpret_cast<const f32x4*>(pr), wb = *reinterpret_cast<const f32x4*>(pr + H), wd = *reinterpret_cast<const f32x4*>(pr + 2 * H);
    f32x4 y;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float fn = nb[k], fd = db[k];
#pragma unroll
        for (int mk = LPR; mk < 64; mk <<= 1) {
            fn += __shfl_xor(fn, mk);
            fd += __shfl_xor(fd, mk);
        }
        const float num = wb[k] + fn, den = wd[k] + fd;
        y[k] = t[k] + num / (den + kAggEps);
    }
    if (group == 0) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sh = *reinterpret_cast<const f32x4*>(shift + c);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(h_in + (int64_t)node * ldh + c);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = relu_keep_nan(y[k] * sc[k] + sh[k]) + hi[k];
        *reinterpret_cast<f32x4*>(h_out + (int64_t)node * H + c) = o;
    }
}

}  // namespace gnnome

extern "C" int gnnome_stream_schedule_sizes(int64_t num_nodes, int64_t num_edges, int rows_per_step, int64_t* steps_capacity_host,
                                            size_t* scratch_bytes_host) {
    using namespace gnnome;
    GN_REQUIRE(steps_capacity_host && scratch_bytes_host, "stream_schedule_sizes: null output");
    GN_REQUIRE(num_nodes >= 0 && num_edges >= 0 && num_nodes < (1ll << 31) && num_edges < (1ll << 31), "stream_schedule_sizes: N / E out of int32 range");
    GN_REQUIRE(rows_per_step == 8 || rows_per_step == 16, "stream_schedule_sizes: rows_per_step must be 8 or 16");
    *steps_capacity_host = num_edges / rows_per_step + num_nodes + 1;
    *scratch_bytes_host = (size_t)num_nodes * (sizeof(int32_t) + 1) + 64;
    return GNNOME_OK;
}

extern "C" int gnnome_build_stream_schedule(int64_t num_nodes, int64_t num_edges, const int32_t* in_ptr, const int32_t* srt_src,
                                            const int32_t* out_ptr, const int32_t* out_pos, const int32_t* out_dst, int num_chunks,
                                            int rows_per_step, int num_slots, int32_t* chunk_node, int32_t* chunk_steps, int32_t* steps,
                                            uint8_t* edge_meta, int32_t* node_pend, int32_t* pend_nodes, int32_t* counters, void* scratch,
                                            size_t scratch_bytes, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(num_nodes > 0 && num_edges > 0 && num_nodes < (1ll << 31) && num_edges < (1ll << 31), "stream_schedule: empty graph or N / E out of range");
    GN_REQUIRE(in_ptr && srt_src && out_ptr && out_pos && out_dst && chunk_node && chunk_steps && steps && edge_meta && node_pend && pend_nodes &&
                   counters && scratch, "stream_schedule: null pointer");
    GN_REQUIRE(num_chunks >= 1 && num_chunks <= (1 << 20), "stream_schedule: num_chunks=%d out of range", num_chunks);
    GN_REQUIRE(rows_per_step == 8 || rows_per_step == 16, "stream_schedule: rows_per_step must be 8 or 16");
    GN_REQUIRE(num_slots >= 1 && num_slots <= 62, "stream_schedule: num_slots=%d not in [1, 62]", num_slots);
    GN_REQUIRE(scratch_bytes >= (size_t)num_nodes * (sizeof(int32_t) + 1), "stream_schedule: scratch too small");
    GN_REQUIRE((uintptr_t)steps % 16 == 0 && (uintptr_t)scratch % 4 == 0, "stream_schedule: steps must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    int32_t* last_near = (int32_t*)scratch;
    uint8_t* slot_of = (uint8_t*)(last_near + num_nodes);
    GN_HIP(hipMemsetAsync(counters, 0, 8 * sizeof(int32_t), s));
    hipLaunchKernelGGL(k_stream_chunk_bounds, dim3((unsigned)((num_chunks + 256) / 256)), dim3(256), 0, s, in_ptr, num_nodes, num_edges, num_chunks, chunk_node);
    hipLaunchKernelGGL(k_stream_sched_nodes, dim3((unsigned)((num_nodes + 255) / 256)), dim3(256), 0, s, num_nodes, num_chunks, (const int32_t*)chunk_node,
                       out_ptr, out_pos, out_dst, edge_meta, last_near, slot_of, node_pend, pend_nodes, counters);
    hipLaunchKernelGGL(k_stream_sched_chunks, dim3((unsigned)((num_chunks + 63) / 64)), dim3(64), 0, s, num_chunks, rows_per_step, num_slots,
                       (const int32_t*)chunk_node, in_ptr, srt_src, (const int32_t*)last_near, slot_of, edge_meta, node_pend, pend_nodes, counters,
                       reinterpret_cast<int4*>(steps), chunk_steps);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

namespace gnnome {
static int launch_stream(const StreamArgs& a, hipStream_t s) {
    constexpr int K = kStreamSlots;
    const int waves = a.H / kStreamCH;
    const size_t lds_bytes = (size_t)waves * (K * 2 * kStreamCH + 4 * 64) * sizeof(float);
    static bool raised = false;
    if (!raised) {
        GN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_aggregate_stream<K, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        GN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_aggregate_stream<K, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised = true;
    }
    // gathers address a node table as a uniform base + a 32-bit byte offset when every table fits 4 GB
    const int64_t ld_max = a.ldn > a.ldh ? (a.ldn > a.H ? a.ldn : a.H) : (a.ldh > a.H ? a.ldh : a.H);
    const bool big = (a.num_nodes * ld_max + 64) * 4 >= (1ll << 32);
#define GN_STREAM(BIG_)                                                                                                                                  \
    hipLaunchKernelGGL((k_aggregate_stream<K, BIG_>), dim3((unsigned)a.chunks), dim3(64 * waves), lds_bytes, s, a.e, a.H, a.A1h, a.A2h, a.A3h, a.ldn, a.h_in, \
                       a.ldh, a.h_out, a.scale, a.shift, a.in_ptr, a.srt_src, a.edge_meta, a.steps, a.chunk_node, a.chunk_steps, a.node_pend, a.pend_rows,     \
                       a.chunks, (int)(a.num_edges - 1))
    if (big) GN_STREAM(true); else GN_STREAM(false);
#undef GN_STREAM
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}
}  // namespace gnnome

extern "C" int gnnome_node_aggregate_stream_f32(const float* e, int hidden, int64_t num_nodes, int64_t num_edges, const float* A1h, const float* A2h, const float* A3h,
                                                int ld_node, const int32_t* in_ptr, const int32_t* srt_src, const int32_t* out_ptr,
                                                const int32_t* out_pos, const int32_t* out_dst, const float* h_in, int ld_h, float* h_out,
                                                const float* norm_scale, const float* norm_shift, int num_chunks, int rows_per_step, int num_slots,
                                                const int32_t* chunk_node, const int32_t* chunk_steps, const int32_t* steps, const uint8_t* edge_meta,
                                                const int32_t* node_pend, const int32_t* pend_nodes, const int32_t* counters, int64_t num_pending,
                                                float* pend_rows, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(num_nodes > 0 && num_edges > 0 && num_edges < (1ll << 31), "node_aggregate_stream: empty graph");
    GN_REQUIRE(e && A1h && A2h && A3h && in_ptr && srt_src && out_ptr && out_pos && out_dst && h_in && h_out && norm_scale && norm_shift && chunk_node &&
                   chunk_steps && steps && edge_meta && node_pend && pend_nodes && counters, "node_aggregate_stream: null pointer");
    GN_REQUIRE(hidden == 64 || hidden == 128 || hidden == 256, "node_aggregate_stream: hidden=%d not in {64,128,256}", hidden);
    GN_REQUIRE(rows_per_step == kStreamRPS, "node_aggregate_stream: the schedule must be built for %d rows per step", kStreamRPS);
    GN_REQUIRE(num_slots >= 1 && num_slots <= 62, "node_aggregate_stream: num_slots=%d not in [1, 62]", num_slots);
    GN_REQUIRE(num_chunks >= 1, "node_aggregate_stream: no chunks");
    GN_REQUIRE(num_pending >= 0 && (num_pending == 0 || pend_rows), "node_aggregate_stream: pending rows missing");
    GN_REQUIRE(ld_node >= hidden && ld_node % 4 == 0 && ld_h >= hidden && ld_h % 4 == 0, "node_aggregate_stream: bad strides");
    GN_REQUIRE(((uintptr_t)A1h % 16 == 0) && ((uintptr_t)A2h % 16 == 0) && ((uintptr_t)A3h % 16 == 0) && ((uintptr_t)h_in % 16 == 0) &&
                   ((uintptr_t)h_out % 16 == 0) && ((uintptr_t)e % 16 == 0) && ((uintptr_t)steps % 16 == 0) &&
                   (pend_rows == nullptr || (uintptr_t)pend_rows % 16 == 0), "node_aggregate_stream: tensors must be 16-byte aligned");
    GN_REQUIRE(h_out != h_in, "node_aggregate_stream: h_out must not alias h_in");
    hipStream_t s = (hipStream_t)stream;
    StreamArgs a;
    a.e = e, a.H = hidden, a.A1h = A1h, a.A2h = A2h, a.A3h = A3h, a.ldn = ld_node, a.h_in = h_in, a.ldh = ld_h, a.h_out = h_out;
    a.scale = norm_scale, a.shift = norm_shift, a.in_ptr = in_ptr, a.srt_src = srt_src, a.edge_meta = edge_meta;
    a.steps = reinterpret_cast<const int4*>(steps), a.chunk_node = chunk_node, a.chunk_steps = chunk_steps, a.node_pend = node_pend;
    a.pend_rows = pend_rows, a.chunks = num_chunks, a.num_nodes = num_nodes, a.num_edges = num_edges;
    // the kernel is built for 62 slots (the LDS a wave owns is a compile-time size); a schedule built for fewer slots runs on it unchanged
    const int rc = launch_stream(a, s);
    if (rc != GNNOME_OK) return rc;
    if (num_pending > 0) {
        const unsigned grid = (unsigned)((num_pending + 3) / 4);
#define GN_FIN(H_)                                                                                                                                 \
    hipLaunchKernelGGL((k_stream_finish<H_>), dim3(grid), dim3(256), 0, s, e, A3h, ld_node, out_ptr, out_pos, out_dst, edge_meta, pend_nodes, counters, \
                       (const float*)pend_rows, h_in, ld_h, h_out, norm_scale, norm_shift)
        if (hidden == 64) GN_FIN(64); else if (hidden == 128) GN_FIN(128); else GN_FIN(256);
#undef GN_FIN
        GN_LAUNCH_CHECK();
    }
    return GNNOME_OK;
}
