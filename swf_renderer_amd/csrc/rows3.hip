// rows3.hip -- the fast row routine of k2_rows (included by raster2.hip; not a translation unit of its own).
//
// One wavefront per (path, <= 64 pixel rows), lane = pixel row, as before -- what changed in round 4 is the work per row:
//   * arithmetic: every quantity of Cairo's scan converter in units of 1 / D, D = 30 (y2 - y1), where it fits 32 bits
//     (FastEdge, device_types.hpp); the one product that does not (A * DX < 2^53) is exact in double precision: an edge's x at a sample
//     row is 2 conversions, 3 f64 operations and two integer fix-ups instead of 64-bit multiplies with fix-up loops; steps are in floor
//     form (add, compare, carry);
//   * gather: lanes = edges work out the rows each staged edge is active in as a 64-bit row mask; a row's active edges are then a BIT
//     MASK built from wave-uniform v_readlane + v_cndmask (no per-lane scan of the edge list, no select chains);
//   * order: the row's (cell, direction, slot) keys are sorted by a NETWORK (19 compare-exchanges for eight edges, min / max on packed
//     keys), the analytic-row test, the tie test and the winding walk then look at neighbours only -- instead of all pairs;
//   * sample rows: the same sort per (row, sample row) lane; a lane walks its sorted cells once and emits Cairo's cells directly; what
//     the classification needs of a row is three bit masks over the path's tile columns (tile columns with a cell, covered in every /
//     in some sample row) combined over a row's sample lanes by DPP rotations -- not a reduction per edge slot;
//   * classification: bit masks over up to 32 tile columns per step instead of a loop over the tile columns.
// A row whose order needs Cairo's list history (two edges on different lines in one cell), or with more than eight active edges,
// or of a chunk with more than STAGE edges in reach, goes to the queue of k2_rows_slow as before.

#if defined(R3_MARKS) && !defined(SWFR_EMU)      // -DR3_MARKS (analysis builds): region markers in the ISA listing
#define R3MARK(n) asm volatile("; R3MARK " #n)
#else
#define R3MARK(n) do { } while (0)
#endif
template <int STAGE> struct R3Mask { typedef uint32_t type; };
template <> struct R3Mask<64> { typedef uint64_t type; };
__device__ __forceinline__ int r3_first(uint32_t m) { return __ffs((int)m) - 1; }
__device__ __forceinline__ int r3_first(uint64_t m) { return __ffsll((long long)m) - 1; }
__device__ __forceinline__ int r3_count(uint32_t m) { return __popc(m); }
__device__ __forceinline__ int r3_count(uint64_t m) { return __popcll(m); }

// bit `lane` of a wave-uniform 64-bit lane mask as 0 / 1: one v_cndmask with the mask as its condition
__device__ __forceinline__ uint32_t lane_bit(uint64_t m, int lane) {
#ifdef SWFR_EMU
    return (uint32_t)((m >> lane) & 1ull);
#else
    (void)lane;
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(r) : "s"(m));
    return r;
#endif
}
// tile columns >= u / tile column t of a block of 32 (any u, t)
__device__ __forceinline__ uint32_t cols_from(int u) { return u >= 32 ? 0u : (u <= 0 ? ~0u : (~0u << u)); }
__device__ __forceinline__ uint32_t col_bit(int t) { return (uint32_t)t < 32u ? (1u << t) : 0u; }

#define R3_BIAS (1 << 24)                                                   // cells are 24.8 positions within +-2^23: biased keys are positive
#define R3_INVALID(s) (0xF0000000u | ((uint32_t)(s) << 5) | (uint32_t)(s))   // sorts behind every cell; distinct per slot; low bits = the slot
#define R3_KEY(c, dir, s) ((((uint32_t)((c) + R3_BIAS)) << 5) | ((dir) > 0 ? 16u : 0u) | (uint32_t)(s))          // (cell, direction, slot): cell < 2^25, slot < 16
// compare-exchange of the sort networks: keys alone, or keys with one payload word
#define R3_CE(i, j) do { const uint32_t lo_ = min(key[i], key[j]), hi_ = max(key[i], key[j]); key[i] = lo_; key[j] = hi_; } while (0)
#define R3_CEP(i, j) do { const bool sw_ = key[i] > key[j]; const uint32_t lo_ = min(key[i], key[j]), hi_ = max(key[i], key[j]); \
                          const int a_ = sw_ ? c1[j] : c1[i], b_ = sw_ ? c1[i] : c1[j]; key[i] = lo_; key[j] = hi_; c1[i] = a_; c1[j] = b_; } while (0)
#define R3_NET4(CE) do { CE(0, 1); CE(2, 3); CE(0, 2); CE(1, 3); CE(1, 2); } while (0)
#define R3_NET6(CE) do { CE(1, 2); CE(4, 5); CE(0, 2); CE(3, 5); CE(0, 1); CE(3, 4); CE(2, 5); CE(0, 3); CE(1, 4); CE(2, 4); CE(1, 3); CE(2, 3); } while (0)
#define R3_NET16(CE) do { CE(0, 1); CE(2, 3); CE(0, 2); CE(1, 3); CE(1, 2); CE(4, 5); CE(6, 7); CE(4, 6); CE(5, 7); CE(5, 6); CE(0, 4); CE(2, 6); CE(2, 4); CE(1, 5); CE(3, 7); CE(3, 5); CE(1, 2); CE(3, 4); CE(5, 6); CE(8, 9); CE(10, 11); CE(8, 10); CE(9, 11); CE(9, 10); CE(12, 13); CE(14, 15); CE(12, 14); CE(13, 15); CE(13, 14); CE(8, 12); CE(10, 14); CE(10, 12); CE(9, 13); CE(11, 15); CE(11, 13); CE(9, 10); CE(11, 12); CE(13, 14); CE(0, 8); CE(4, 12); CE(4, 8); CE(2, 10); CE(6, 14); CE(6, 10); CE(2, 4); CE(6, 8); CE(10, 12); CE(1, 9); CE(5, 13); CE(5, 9); CE(3, 11); CE(7, 15); CE(7, 11); CE(3, 5); CE(7, 9); CE(11, 13); CE(1, 2); CE(3, 4); CE(5, 6); CE(7, 8); CE(9, 10); CE(11, 12); CE(13, 14); } while (0)     /* Batcher's odd-even merge sort: 63 compare-exchanges */
#define R3_NET8(CE) do { CE(0, 1); CE(2, 3); CE(4, 5); CE(6, 7); CE(0, 2); CE(1, 3); CE(4, 6); CE(5, 7); CE(1, 2); CE(5, 6); CE(0, 4); CE(3, 7); \
                         CE(1, 5); CE(2, 6); CE(1, 4); CE(3, 6); CE(2, 4); CE(3, 5); CE(3, 4); } while (0)

// Cells of a boundary edge of an analytically converted row (A.5 render_edge): qt / qb = x (24.8, quotient) at the row's top and
// bottom, rl = the remainder (units of 1 / D) of the LEFT one of the two; writes exactly full_span(qt, qb) cells.  The row's x extent
// in units of 1 / D is W = 512 |DX| in every row the edge crosses completely, the first column's share of the fifteen sample rows
// floor(((X - x_left) D) / W) with X the column's right side; products below 2^53: exact in double precision.
__device__ __forceinline__ void full_cells3(int32_t qt, int32_t qb, int32_t rl, int32_t DX, int32_t D, double invW, int32_t fq0, double fr0, int neg,
                                            int xminp, int xmaxp, Cell* __restrict__ dst) {
    // neg: 0 for a left edge (+ heights), -1 for a right edge: (h ^ neg) - neg negates without a multiply
    int ix1 = qt >> 8, f1 = qt & 255, ix2 = qb >> 8, f2 = qb & 255;
    if (ix1 == ix2) { put_cell_h(dst, ix1, (15 ^ neg) - neg, f1 + f2, xminp, xmaxp); return; }
    int32_t ql = qt;
    if (ix2 < ix1) { int t = ix1; ix1 = ix2; ix2 = t; t = f1; f1 = f2; f2 = t; ql = qb; }
    const int span = ix2 - ix1 + 1;
    const double W = 512.0 * fabs((double)DX);
    const double num = fma((double)((ix1 + 1) * 256 - ql), (double)D, -(double)rl);
    double yqf = floor(num * invW);
    double yr = fma(-yqf, W, num);
    if (yr < 0.0) { yqf -= 1.0; yr += W; }
    if (yr >= W) { yqf += 1.0; yr -= W; }
    int yq = (int)yqf;
    int y_prev = yq;
    if (span <= MAX_CELLS_PER_EDGE_ROW) {
        put_cell_h(dst, ix1, (y_prev ^ neg) - neg, 256 + f1, xminp, xmaxp);
#pragma unroll 1
        for (int k = 1; k < span - 1; ++k) {
            yq += fq0; yr += fr0; if (yr >= W) { ++yq; yr -= W; }
            put_cell_h(dst + k, ix1 + k, ((yq - y_prev) ^ neg) - neg, 256, xminp, xmaxp);
            y_prev = yq;
        }
        put_cell_h(dst + span - 1, ix2, ((15 - y_prev) ^ neg) - neg, f2, xminp, xmaxp);
        return;
    }
    // an edge over more columns than that: at most 17 of its cells have a height (they add up to the fifteen sample rows)
    int n = 0;
    if (y_prev) put_cell_h(dst + n++, ix1, (y_prev ^ neg) - neg, 256 + f1, xminp, xmaxp);
#pragma unroll 1
    for (int c = ix1 + 1; c < ix2; ++c) {
        yq += fq0; yr += fr0; if (yr >= W) { ++yq; yr -= W; }
        const int h = yq - y_prev;
        if (h && n < MAX_CELLS_PER_EDGE_ROW - 1) put_cell_h(dst + n++, c, (h ^ neg) - neg, 256, xminp, xmaxp);
        y_prev = yq;
    }
    if (15 - y_prev) put_cell_h(dst + n++, ix2, ((15 - y_prev) ^ neg) - neg, f2, xminp, xmaxp);
    while (n < MAX_CELLS_PER_EDGE_ROW) put_cell(dst + n++, xminp, 0, 0, xminp, xmaxp);
}

// STAGE: edges of the path a chunk keeps in LDS (two instances of the kernel: 32 for scenes whose paths have at most 32 edges, 64
// for the others; a chunk with more edges in reach leaves its rows to the queue)
template <int STAGE, int NS>
__device__ __forceinline__ void rows3_chunk_body(FramePtr FR, uint32_t block) {
    typedef typename R3Mask<STAGE>::type amask_t;
    __shared__ __attribute__((aligned(16))) FastEdge staged[STAGE];
    __shared__ uint32_t sub_cells[4][NS * 15];        // the cells of the four rows of a sample pass, before they are copied out coalesced
    __shared__ uint32_t sub_masks[4][4];                       // ... and the rows' tile-column masks: cells, covered in all / in some sample rows
    __shared__ uint8_t slot_role[NS][64];             // role per edge slot, scattered there from the sorted order (column = lane: private)
    __shared__ uint8_t slot_edge[NS][64];             // staged edge per slot (the tie check looks edges up by sorted position)
    __shared__ uint32_t mid_bits[2];                           // rows of the chunk in which a staged edge starts or ends
    TRACE_DECL;
    TRACE_NOWAIT(0);
    const int lane = threadIdx.x;
    const ChunkInfo ck = FR->chunks[block];                               // wave-uniform: path and edge reads are scalar
    const uint32_t lo = ck.path;
    // the first sixty-four edges of the path are requested at once -- the chunk record carries the path's edge range -- so that they
    // travel beside the path record and the band slots instead of behind them
    const FastEdge* __restrict__ FE = fast_edges_of(FR->edges, FR->n_edges) + ck.first_edge;
    const FastEdge ek_first = FE[min((uint32_t)lane, ck.n_edges ? ck.n_edges - 1u : 0u)];
    const DevPath P = FR->paths[lo];
    const int r = (int)ck.first_row + lane;
    const int chunk_rows = (int)ck.rows;                                 // 16, 32 or 64: whole tile-rows, starting on a tile-row boundary -- or 8: one strip's rows (a frame of few, tall paths)

    // the band entry of this lane's tile-row: where its row headers and class bytes go
    const int g16 = (((int)ck.first_row & (TILE_H - 1)) + lane) >> 4;     // the lane's tile-row, counted from the chunk's first (an 8-row chunk may start in the middle of one)
    const int band = (int)ck.first_row / TILE_H + g16;
    const int band_lo = P.y_min / TILE_H, band_hi = (P.y_max - 1) / TILE_H;
    // (the band records are requested with what the CHUNK record says -- clamped where the path may turn out not to reach -- so that
    //  they travel beside the path record, not behind it; band_ok decides afterwards whether they mean anything)
    BandSlot cls_bs = {0u, 0u, 0u, 0u};
    uint32_t cls_b0 = 0, cls_b1 = 0;
    if (ck.slot0 != ~0u && lane < chunk_rows) {
        cls_bs = FR->band_slots[ck.slot0 + (uint32_t)g16];                 // (at most three records past the path's own: the host reserves eight spare ones)
        const uint32_t bclamp = min((uint32_t)band, FR->n_bands - 1u);
        cls_b0 = FR->band_off[bclamp];
        cls_b1 = FR->band_off[bclamp + 1u];
    }
    const bool band_ok = ck.slot0 != ~0u && lane < chunk_rows && band >= band_lo && band <= band_hi && P.kind == SWFR_PATH_TOR;
    const uint32_t ri = band_ok ? cls_bs.slot * TILE_H + (uint32_t)(r & (TILE_H - 1)) : ~0u;
    const bool in_path = P.kind == SWFR_PATH_TOR && lane < chunk_rows && r >= P.y_min && r < P.y_max;
    bool live = in_path;
    { uint32_t lb; if (live && !owns_band(FR, r / TILE_H, lb)) live = false; }         // another rank's tile-row
    if (__ballot(live) == 0ull) {
        // nothing of this chunk is this handle's (multi-GPU): its rows stay "not known here" for the slow rows' history look-ups
        if (ri != ~0u && lane < chunk_rows) { RowInfo2 h; h.off = 0; h.n = 0; h.mode = (uint16_t)(in_path ? (uint32_t)ROW_FOREIGN : (uint32_t)ROW_EMPTY); FR->rows[ri] = h; }
        return;
    }
    TRACE(1);                                                            // chunk, path and band records in
    R3MARK(1);
    const int fast_limit = min((int)FR->fast_limit, NS);
    // ---- stage the edges that can be active in this chunk's rows (path order kept)
    const int lo_s = (int)ck.first_row * 15, hi_s = lo_s + chunk_rows * 15;
    if (lane < 2) mid_bits[lane] = 0u;
    uint32_t n_list = 0;
    bool use_lds = true;
    int inc_before = 0;                                                  // (edge, pixel row) pairs of this path above the chunk: where its cells start
    for (uint32_t eb = 0; eb < ck.n_edges; eb += 64) {
        const uint32_t k = eb + (uint32_t)lane;
        const FastEdge ek = eb == 0 ? ek_first : FE[min(k, ck.n_edges - 1u)];
        const bool valid = k < ck.n_edges && ek.ybot > ek.ytop;
        if (valid) inc_before += max(0, min((ek.ybot - 1) / 15 + 1, (int)ck.first_row) - ek.ytop / 15);
        const bool hit = use_lds && valid && ek.ytop < hi_s && ek.ybot > lo_s;
        const unsigned long long hb = __ballot(hit);
        const uint32_t at = n_list + (uint32_t)__popcll(hb & ((1ull << lane) - 1ull));
        if (hit && at < (uint32_t)STAGE) staged[at] = ek;
        n_list += (uint32_t)__popcll(hb);
        if (n_list > (uint32_t)STAGE) use_lds = false;                        // (the loop goes on: every edge of the path counts for inc_before)
    }
    const uint32_t chunk_cell_base = (ck.rec_base + (uint32_t)__builtin_amdgcn_readlane(wave_scan_incl(inc_before), 63)) * (uint32_t)MAX_CELLS_PER_EDGE_ROW;
    // More edges in reach than the staging area holds (a path of hundreds of edges): the rows with more than fast_limit active edges
    // go to the queue anyway -- stage again, only the edges of the OTHER rows; if even those are too many, every row with an edge is queued.
    bool forced_over = false;                                            // this lane's row is left to the queue whatever the staged edges say
    if (!use_lds) {                                                      // (wave-uniform)
        int cnt = 0;
        for (uint32_t k = 0; k < P.n_edges; ++k) {                        // (wave-uniform: scalar loads)
            const int yt = FE[k].ytop, yb = FE[k].ybot;
            if (yb > r * 15 && yt < r * 15 + 15) ++cnt;
        }
        const bool fast_row = live && cnt > 0 && cnt <= fast_limit;
        const unsigned long long fm = __ballot(fast_row);
        forced_over = live && cnt > 0 && !fast_row;
        n_list = 0;
        bool ok = fm != 0ull;
        lds_barrier();                                                   // (the staging area is rewritten)
        for (uint32_t eb = 0; ok && eb < P.n_edges; eb += 64) {
            const uint32_t k = eb + (uint32_t)lane;
            const FastEdge ek = FE[min(k, P.n_edges - 1u)];
            const int ra = ek.ytop / 15 - (int)ck.first_row, rz = (ek.ybot + 14) / 15 - (int)ck.first_row;
            const int a = min(max(ra, 0), 64), b = min(max(rz, 0), 64);
            const uint64_t bits = (b >= 64 ? ~0ull : ((1ull << b) - 1ull)) & ~(a >= 64 ? ~0ull : ((1ull << a) - 1ull));
            const bool hit = k < P.n_edges && ek.ybot > ek.ytop && (bits & fm) != 0ull;
            const unsigned long long hb = __ballot(hit);
            const uint32_t at = n_list + (uint32_t)__popcll(hb & ((1ull << lane) - 1ull));
            if (hit && at < (uint32_t)STAGE) staged[at] = ek;
            n_list += (uint32_t)__popcll(hb);
            if (n_list > (uint32_t)STAGE) ok = false;
        }
        use_lds = ok;
        if (!ok) { n_list = 0; forced_over = live && cnt > 0; }
    }
    lds_barrier();
    R3MARK(2);
    // ---- lanes = staged edges: the chunk's rows an edge is active in, as a bit mask over the rows; rows it starts or ends inside
    uint32_t rb_lo = 0, rb_hi = 0;
    if (use_lds && (uint32_t)lane < n_list) {
        const int yt = staged[lane].ytop, yb = staged[lane].ybot;            // 0 <= yt < yb
        const int ra = yt / 15 - (int)ck.first_row, rz = (yb + 14) / 15 - (int)ck.first_row;      // pixel rows [ra, rz) relative to the chunk
        const int a = min(max(ra, 0), 64), b = min(max(rz, 0), 64);
        const uint64_t below_b = b >= 64 ? ~0ull : ((1ull << b) - 1ull), below_a = a >= 64 ? ~0ull : ((1ull << a) - 1ull);
        const uint64_t bits = below_b & ~below_a;
        rb_lo = (uint32_t)bits; rb_hi = (uint32_t)(bits >> 32);
        uint64_t mb = 0;
        if (yt % 15 != 0 && ra >= 0 && ra < 64) mb |= 1ull << ra;
        if (yb % 15 != 0 && rz >= 1 && rz <= 64) mb |= 1ull << (rz - 1);
        if ((uint32_t)mb) atomicOr(&mid_bits[0], (uint32_t)mb);
        if ((uint32_t)(mb >> 32)) atomicOr(&mid_bits[1], (uint32_t)(mb >> 32));
    }
    lds_barrier();
    TRACE(2);                                                            // edges staged, their row masks made
    R3MARK(3);
    // ---- lanes = rows: the row's active edges as a bit mask over the staged edges (bit k from edge k's row mask: wave-uniform)
    amask_t amask = 0;
    if (use_lds)
        for (int k = (int)n_list - 1; k >= 0; --k) {                         // wave-uniform
            const uint64_t m = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)rb_hi, k) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)rb_lo, k);
            amask = (amask_t)(amask << 1) | (amask_t)lane_bit(m, lane);
        }
    const uint64_t mid_mask = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)mid_bits[1]) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)mid_bits[0]);
    const bool mid_row = lane_bit(mid_mask, lane) != 0u;
    if (!live) amask = 0;
    int n = r3_count(amask);
    const bool overflow = live && (forced_over || n > fast_limit);
    if (overflow) { n = 0; amask = 0; }
    // wave-uniform bound on the active edges of any row of this wave: the unrolled slot loops stop there
    const int nmax = (NS > 8 && __ballot(n > 8)) ? 16 : __ballot(n > 6) ? 8 : __ballot(n > 4) ? 6 : __ballot(n > 2) ? 4 : 2;
    const int s0 = r * 15;
    const unsigned fmask = P.fill_rule ? 1u : ~0u;

    R3MARK(4);
    // ---- rows that can be converted analytically: x of every active edge at the first sample row of this pixel row and of the next
    uint32_t key[NS]; int c1[NS];
    int32_t qt[NS], qb[NS], rl[NS];                    // (a slot's staged edge is looked up in slot_edge[], its role in `roles`: registers are this kernel's occupancy)
#ifdef ABL3_NOEVAL
    const bool can_full = false;
#else
    const bool can_full = n > 0 && !mid_row;
#endif
    {
        amask_t m = amask;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            key[s] = R3_INVALID(s); c1[s] = 0x7fffffff; qt[s] = qb[s] = 0; rl[s] = 0;
            if (s >= nmax) continue;                          // wave-uniform
            const int k = m ? r3_first(m) : 0;                // (no edge left: the first staged record, the results are not used)
            m &= m - 1;
            slot_edge[s][lane] = (uint8_t)k;
            const FastEdge& e = staged[k];
            const int32_t x1 = e.x1, DX = e.DX, D = e.D, hD = D >> 1;
            int32_t q, rm;
            fast_x_at(e.a0, DX, D, e.invD, s0, q, rm);
            const int32_t xa = x1 + q;
            const int c0 = xa + (rm >= hD ? 1 : 0);
            // fifteen sample rows on: + 7680 DX / D in floor form -- the same unique (quotient, remainder in [0, D)) as the closed form
            int32_t xb = xa + e.q15, rn = rm + e.r15;
            if (rn >= D) { ++xb; rn -= D; }
            const int c1v = xb + (rn >= hD ? 1 : 0);
            // half a sample row back: the row's top and bottom
            const int32_t hq = e.hq, hr = e.hr;
            int32_t qa = xa - hq, ra = rm - hr; if (ra < 0) { --qa; ra += D; }
            int32_t qz = xb - hq, rz = rn - hr; if (rz < 0) { --qz; rz += D; }
            const bool valid = can_full && s < n;
            if (valid) { key[s] = R3_KEY(c0, e.dir, s); c1[s] = c1v; }
            qt[s] = qa; qb[s] = qz; rl[s] = DX < 0 ? rz : ra;
        }
    }
    TRACE(3);                                                            // gathered and evaluated
    R3MARK(5);
    // ---- sort by (cell, slot); the analytic test, ties and the winding walk then look at neighbours only
    if (nmax <= 2) R3_CEP(0, 1);
    else if (nmax <= 4) R3_NET4(R3_CEP);
    else if (nmax <= 6) R3_NET6(R3_CEP);
    else if (nmax <= 8) R3_NET8(R3_CEP);
    else if constexpr (NS > 8) R3_NET16(R3_CEP);
    bool full = can_full, deep = false;
    unsigned tie_bits = 0;                                     // bit p: the edges at sorted positions p and p + 1 share a cell
#pragma unroll
    for (int p = 0; p + 1 < NS; ++p) {
        if (p + 1 >= nmax) continue;                          // wave-uniform
        if (c1[p] > c1[p + 1]) full = false;                  // (slots without an edge sort last with the largest c1: never a violation)
        if ((key[p] >> 5) == (key[p + 1] >> 5)) { tie_bits |= 1u << p; deep = true; }
    }
    // coincident cells: edges on one and the same line (a shape edge with fill0 == fill1 is there twice) can go in either order -- slot
    // order is used; any other tie needs the history of Cairo's edge list: the slow-row kernel's job
    bool defer = false;
    if (__ballot(deep && can_full) != 0ull) {
        if (deep && can_full) {
            bool real = false;
#pragma unroll
            for (int p = 0; p + 1 < NS; ++p) {
                if (p + 1 >= nmax) continue;
                if ((tie_bits >> p) & 1u) {
                    const FastEdge& ea = staged[slot_edge[key[p] & 15u][lane]];
                    const FastEdge& eb = staged[slot_edge[key[p + 1] & 15u][lane]];
                    real |= !(ea.x1 == eb.x1 && ea.a0 == eb.a0 && ea.DX == eb.DX && ea.D == eb.D);       // same_line
                }
            }
            defer = real;
        }
    }
    uint32_t mode = ROW_EMPTY;
    bool is_sub = false;
    uint32_t roles = 0;                                        // two bits per slot: 0 no boundary, 1 left edge of a span, 2 right edge
#define R3_ROLE(s) ((roles >> (2 * (s))) & 3u)
    if (n > 0) {
        if (defer) mode = ROW_DEFER;
        else if (full) mode = ROW_FULL;
        else { mode = ROW_SUB; is_sub = true; }
    }
    if (__ballot(mode == ROW_FULL) != 0ull) {
        // winding walk over the sorted edges; the roles go back to the edges' slots through the lane's own LDS column
        int w = 0;
#pragma unroll
        for (int p = 0; p < NS; ++p) {
            if (p >= nmax) continue;                          // wave-uniform
            const bool in_b = ((unsigned)w & fmask) != 0;
            w += (key[p] & 16u) ? 1 : -1;
            const bool in_a = ((unsigned)w & fmask) != 0;
            const bool fg = p == 0 || !((tie_bits >> (p - 1)) & 1u), lg = !((tie_bits >> p) & 1u);
            uint32_t ro = 0;
            if (!in_b && fg) ro = 1u;                         // left edge of a span
            else if (!in_a && lg) ro = 2u;                    // right edge
            slot_role[key[p] & 15u][lane] = (uint8_t)ro;
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s >= nmax) continue;
            if (mode == ROW_FULL && s < n) roles |= (uint32_t)slot_role[s][lane] << (2 * s);
        }
    }
    R3MARK(6);
    // ---- room for the rows' cells inside the wavefront's region: a FULL row takes the exact number of its cells, a SUB row room for
    //      one cell per (active edge, sample row) -- its cells are counted while they are made, so the row uses a prefix of its room
    int n_cells = 0;
    if (mode == ROW_FULL && ri != ~0u) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s >= nmax) continue;                          // wave-uniform
            if (R3_ROLE(s) != 0) n_cells += full_span(qt[s], qb[s]);
        }
    }
    const int room = is_sub && ri != ~0u ? n * 15 : n_cells;
    const uint32_t incl_cells = (uint32_t)wave_scan_incl(room);
    const uint32_t total_cells = (uint32_t)__builtin_amdgcn_readlane((int)incl_cells, 63);
    // the wavefront's cells start at its chunk's slot: (edge, row) pairs before it x MAX_CELLS_PER_EDGE_ROW -- no allocator in this kernel
    const uint32_t wave_base = ((uint64_t)chunk_cell_base + total_cells <= (uint64_t)FR->cell_slice) ? chunk_cell_base : ~0u;
    if (wave_base == ~0u && lane == 0) atomicOr(&FR->counters[C2_ERROR], E2_CELL_ARENA);
    const uint32_t my_room = wave_base + incl_cells - (uint32_t)room;
    R3MARK(7);
    // ---- FULL rows: the cells of every boundary edge, from the exact end points, into the row's room
#ifdef ABL3_NOCELLS
    if (ck.slot0 == 0x7ffffff0u)
#endif
    if (mode == ROW_FULL && ri != ~0u && wave_base != ~0u) {
        uint32_t off = my_room;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s >= nmax) continue;
            if (R3_ROLE(s) != 0) {
                const FastEdge& e = staged[slot_edge[s][lane]];
                full_cells3(qt[s], qb[s], rl[s], e.DX, e.D, e.invW, e.fq, e.fr, (R3_ROLE(s) & 1u) ? 0 : -1, P.x_min, P.x_max, &FR->cells[off]);
                off += (uint32_t)full_span(qt[s], qb[s]);
            }
        }
    }
    R3MARK(8);
    // ---- tile-column masks of the row over the first 32 tile columns of the path's rectangle (classification below): columns with a
    //      cell, columns covered in every / in some sample row
    const int tc0 = P.x_min / TILE_W, tc1 = (P.x_max - 1) / TILE_W, ntc = tc1 - tc0 + 1;
    uint32_t m_inter = 0, m_and = 0, m_or = 0;
    auto full_row_masks = [&](int tbase, uint32_t& inter, uint32_t& cov) {
        // a boundary edge's cells lie in the pixel columns [clo, chi]; everything right of it changes sides
        inter = 0; cov = 0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s >= nmax) continue;
            if (R3_ROLE(s) != 0) {
                const int a = qt[s] >> 8, b = qb[s] >> 8;
                const int tlo = (min(a, b) >> 6) - tbase, thi = (max(a, b) >> 6) - tbase;          // (arithmetic shifts: floor)
                inter |= cols_from(tlo) & ~cols_from(thi + 1);
                cov ^= cols_from(thi + 1);
            }
        }
    };
    if (mode == ROW_FULL) { uint32_t iv, cv; full_row_masks(tc0, iv, cv); m_inter = iv; m_and = cv; m_or = cv; }
    lds_barrier();
    TRACE(4);                                                            // sorted, roles, FULL cells and masks
    R3MARK(9);
    // ---- the wave's SUB rows, 4 rows per pass: lanes 16g .. 16g + 14 are the fifteen sample rows of the pass's g-th row (lane 16g + 15
    //      idles): a lane sorts its sample row's cells, walks them once and emits Cairo's cells (A.5 add_subspan); a row's lanes are one
    //      DPP row, so the cell counts and the classification masks combine by row shifts / rotations: no LDS atomics
    unsigned long long pending = __ballot(is_sub);
#ifdef ABL3_NOSUB
    pending = 0ull;
#endif
    const int g = lane >> 4, sub = lane & 15;
    while (pending) {
        unsigned long long m = pending;
        int R = -1;
        for (int t = 0; t <= g; ++t) { if (!m) { R = -1; break; } R = __ffsll((long long)m) - 1; m &= m - 1; }
        const unsigned long long pass_rows = pending;           // its four lowest bits set are this pass's rows
        for (int t = 0; t < 4 && pending; ++t) pending &= pending - 1;
        // cross-lane reads must run with every lane active: ds_bpermute returns 0 for a disabled source lane
        const int Rsrc = R >= 0 ? R : 0;
        const int nR = __shfl(n, Rsrc);
        const int rR = __shfl(r, Rsrc);
        const uint32_t riR = (uint32_t)__shfl((int)ri, Rsrc);
        amask_t mR;
        if (sizeof(amask_t) == 8) mR = (amask_t)(((uint64_t)(uint32_t)__shfl((int)(uint32_t)((uint64_t)amask >> 32), Rsrc) << 32) | (uint64_t)(uint32_t)__shfl((int)(uint32_t)amask, Rsrc));
        else mR = (amask_t)(uint32_t)__shfl((int)(uint32_t)amask, Rsrc);
        // the slot loops of this pass stop at the most active edges of ITS rows (wave-uniform)
        int nmaxp = 0;
        {
            unsigned long long m2 = pass_rows;
            for (int t = 0; t < 4 && m2; ++t) { nmaxp = max(nmaxp, __builtin_amdgcn_readlane(n, __ffsll((long long)m2) - 1)); m2 &= m2 - 1; }
            nmaxp = nmaxp > 8 ? 16 : nmaxp > 6 ? 8 : nmaxp > 4 ? 6 : nmaxp > 2 ? 4 : 2;
        }
        const bool sampling = R >= 0 && sub < 15;
        const int ss = rR * 15 + sub;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            key[s] = R3_INVALID(s);
            if (s >= nmaxp) continue;                            // wave-uniform
            const int k = mR ? r3_first(mR) : 0;
            mR &= mR - 1;
            const FastEdge& e = staged[k];
            int32_t q, rm;
            fast_x_at(e.a0, e.DX, e.D, e.invD, ss, q, rm);
            const int c = e.x1 + q + (rm >= (e.D >> 1) ? 1 : 0);
            if (sampling && s < nR && e.ytop <= ss && ss < e.ybot) key[s] = R3_KEY(c, e.dir, s);
        }
        if (nmaxp <= 2) R3_CE(0, 1);
        else if (nmaxp <= 4) R3_NET4(R3_CE);
        else if (nmaxp <= 6) R3_NET6(R3_CE);
        else if (nmaxp <= 8) R3_NET8(R3_CE);
        else if constexpr (NS > 8) R3_NET16(R3_CE);
        // walk: a cell where the inside state differs before and after a group of edges in one cell
        uint32_t cw[NS];
        unsigned em = 0;
        uint32_t s_inter = 0, s_cov = 0;
        {
            int w = 0; bool in_prev = false;
#pragma unroll
            for (int p = 0; p < NS; ++p) {
                cw[p] = 0;
                if (p >= nmaxp) continue;                        // wave-uniform
                const bool valid = key[p] < 0xF0000000u;
                const uint32_t cellp = key[p] >> 5, celln = key[p + 1 < NS ? p + 1 : p] >> 5;
                if (valid) w += (key[p] & 16u) ? 1 : -1;
                const bool last = p + 1 >= NS || cellp != celln;
                const bool in_now = ((unsigned)w & fmask) != 0;
                const bool emit = valid && last && in_now != in_prev;
                if (valid && last) in_prev = in_now;
                const int c = (int)cellp - R3_BIAS;
                cw[p] = pack_sub_cell(c, in_now ? 1 : -1, P.x_min, P.x_max);
                if (emit) {
                    em |= 1u << p;
                    const int t = (c >> 14) - tc0;                // its tile column: everything right of it changes sides
                    s_inter |= col_bit(t);
                    s_cov ^= cols_from(t + 1);
                }
            }
        }
        if (ntc > 32) s_inter = ~0u;                            // (a path wider than 32 tile columns: its sampled rows count as partial everywhere)
        // cells of this lane, of the row's lanes before it, of the row
        const bool keep = riR != ~0u;
        const int cnt = keep ? __popc(em) : 0;
        int sc = cnt;
        sc += __builtin_amdgcn_update_dpp(0, sc, 0x111, 0xf, 0xf, false);   // row_shr:1 .. 8: inclusive scan over the row's sixteen lanes
        sc += __builtin_amdgcn_update_dpp(0, sc, 0x112, 0xf, 0xf, false);
        sc += __builtin_amdgcn_update_dpp(0, sc, 0x114, 0xf, 0xf, false);
        sc += __builtin_amdgcn_update_dpp(0, sc, 0x118, 0xf, 0xf, false);
        {
            uint32_t at = (uint32_t)(sc - cnt);
#pragma unroll
            for (int p = 0; p < NS; ++p) {
                if (p >= nmaxp) continue;
                if (keep && ((em >> p) & 1u)) sub_cells[g][at++] = cw[p];
            }
        }
        // all-reduce of the masks over the row's sixteen lanes (four rotations each); the idle lanes hold the identities
        uint32_t a_and = sampling ? s_cov : ~0u, a_or = sampling ? s_cov : 0u, a_int = sampling ? s_inter : 0u;
        a_and &= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a_and, 0x128, 0xf, 0xf, false);
        a_or |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a_or, 0x128, 0xf, 0xf, false);
        a_int |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a_int, 0x128, 0xf, 0xf, false);
        a_and &= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a_and, 0x124, 0xf, 0xf, false);
        a_or |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a_or, 0x124, 0xf, 0xf, false);
        a_int |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a_int, 0x124, 0xf, 0xf, false);
        a_and &= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a_and, 0x122, 0xf, 0xf, false);
        a_or |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a_or, 0x122, 0xf, 0xf, false);
        a_int |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a_int, 0x122, 0xf, 0xf, false);
        a_and &= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a_and, 0x121, 0xf, 0xf, false);
        a_or |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a_or, 0x121, 0xf, 0xf, false);
        a_int |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a_int, 0x121, 0xf, 0xf, false);
        if (sub == 0 && R >= 0) { sub_masks[g][0] = a_int; sub_masks[g][1] = a_and; sub_masks[g][2] = a_or; }
        lds_barrier();                                          // the pass's cells and masks are staged
        {   // the row lanes of this pass take their masks over (my_t: the lane's place among the pass's four rows)
            const int my_t = (int)__popcll(pass_rows & ((1ull << lane) - 1ull));
            if (((pass_rows >> lane) & 1ull) != 0ull && my_t < 4) { m_inter = sub_masks[my_t][0]; m_and = sub_masks[my_t][1]; m_or = sub_masks[my_t][2]; }
        }
        // ---- copy out (coalesced) into the rows' room, row headers
        {
            const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane(sc, 15), c1r = (uint32_t)__builtin_amdgcn_readlane(sc, 31);
            const uint32_t c2 = (uint32_t)__builtin_amdgcn_readlane(sc, 47), c3 = (uint32_t)__builtin_amdgcn_readlane(sc, 63);
            const uint32_t total = c0 + c1r + c2 + c3;
            // where the rooms of the pass's rows start: the row lanes know (exclusive prefix of the rooms), every lane asks
            int Rg[4];
            {
                unsigned long long m2 = pass_rows;
#pragma unroll
                for (int t = 0; t < 4; ++t) { Rg[t] = m2 ? __ffsll((long long)m2) - 1 : 0; m2 &= m2 - 1; }
            }
            const uint32_t b0 = (uint32_t)__shfl((int)my_room, Rg[0]), b1 = (uint32_t)__shfl((int)my_room, Rg[1]);
            const uint32_t b2 = (uint32_t)__shfl((int)my_room, Rg[2]), b3 = (uint32_t)__shfl((int)my_room, Rg[3]);
            if (wave_base != ~0u) {
                for (uint32_t t = (uint32_t)lane; t < total; t += 64) {
                    const int gg = t < c0 ? 0 : (t < c0 + c1r ? 1 : (t < c0 + c1r + c2 ? 2 : 3));
                    const uint32_t pre = gg == 0 ? 0u : (gg == 1 ? c0 : (gg == 2 ? c0 + c1r : c0 + c1r + c2));
                    const uint32_t bb = gg == 0 ? b0 : (gg == 1 ? b1 : (gg == 2 ? b2 : b3));
                    FR->cells[bb + (t - pre)] = Cell{sub_cells[gg][t - pre]};
                }
            }
            if (sub == 0 && R >= 0 && riR != ~0u) {          // the first sample lane of each of the pass's rows writes its header
                const uint32_t bb = g == 0 ? b0 : (g == 1 ? b1 : (g == 2 ? b2 : b3));
                const uint32_t cg = g == 0 ? c0 : (g == 1 ? c1r : (g == 2 ? c2 : c3));
                RowInfo2 h; h.off = wave_base == ~0u ? 0u : bb; h.n = wave_base == ~0u ? (uint16_t)0 : (uint16_t)cg; h.mode = (uint16_t)ROW_SUB;
                FR->rows[riR] = h;
            }
        }
        lds_barrier();                                          // the staging has been read: the next pass may overwrite it
    }
    TRACE(5);                                                            // sample passes
    R3MARK(10);
    const bool slow = live && (overflow || defer);
    // ---- headers of the rows that are not SUB (those were written with their cells); a FULL row's cells are in place already
    if (ri != ~0u && lane < chunk_rows && mode != ROW_SUB) {
        RowInfo2 h; h.off = 0; h.n = 0; h.mode = (uint16_t)(slow ? (uint32_t)ROW_DEFER : (in_path && !live) ? (uint32_t)ROW_FOREIGN : mode);   // (another rank's row: not known here)
        if (mode == ROW_FULL && !slow && wave_base != ~0u) { h.off = my_room; h.n = (uint16_t)n_cells; }
        FR->rows[ri] = h;
    }
    // ---- rows left to the slow-row kernel
    {
        const bool q = slow && ri != ~0u;
        const unsigned long long qm = __ballot(q);
        if (qm) {
            uint32_t qbase = 0;
            if (lane == 0) {
                qbase = atomicAdd(&FR->counters[C2_SLOW], (uint32_t)__popcll(qm));
                if (atomicOr(&FR->path_flag[lo], 1u) == 0u) FR->path_queue[atomicAdd(&FR->counters[C2_PATHQ], 1u)] = lo;   // once per path
            }
            qbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)qbase);
            if (q) {
                const uint32_t at = qbase + (uint32_t)__popcll(qm & ((1ull << lane) - 1ull));
                if (at < FR->slow_cap) {
                    SlowRow sr; sr.path = lo; sr.row = r; sr.ri = ri;
                    // where the path's (path, tile-row) pairs start in band_slots (the slow kernel looks up earlier rows' headers) | tie flag
                    sr.pad = (ck.slot0 - (uint32_t)((int)ck.first_row / TILE_H - band_lo)) | (defer ? 0x80000000u : 0u);
                    FR->slow[at] = sr;
                }
                else atomicOr(&FR->counters[C2_ERROR], E2_SLOW_QUEUE);
            }
        }
    }
    R3MARK(11);
    // ---- classification of this chunk's (strip, tile column, path) triples from the rows' masks: lanes 8j .. 8j + 7 are the pixel rows
    //      of the chunk's j-th strip.  Only the columns of the path's rectangle are written (the rest of the class matrix was cleared
    //      when the scene was uploaded and nothing ever writes there).  32 tile columns per step.
#ifdef ABL3_NOCLASS
    if (ck.slot0 == 0x7ffffff0u)
#endif
    if (ck.slot0 != ~0u && P.kind == SWFR_PATH_TOR) {
        const int width = FR->width, height = FR->height;
        uint8_t* out = FR->cls;
        uint32_t n_b = 0;
        if (band_ok) {
            n_b = cls_b1 - cls_b0;
            out = FR->cls + (size_t)STRIPS_PER_TILE * FR->tiles_x * cls_b0 + (cls_bs.slot - cls_b0);
        }
        const swfr_style& st = style_at(FR, P.style);           // (kind and pixel only)
        const uint32_t opq = (st.kind == SWFR_STYLE_SOLID && P.lerp && (st.pixel >> 24) == 0xffu) ? CLS_OPAQUE : 0u;
        const bool in_frame = r < height && band_ok, in_rows = in_frame && in_path;
        uint32_t local_trow = 0;
        const bool own_band = band_ok && owns_band(FR, band, local_trow);
        const bool cost_order = FR->strip_order != 0u;
        const uint32_t pos1 = cls_bs.slot - cls_b0 + 1u;         // the path's position in its tile-row's band list, 1-based
        const int strip_in_tile = (r >> 3) & 1, tsub = lane & 7;          // (a chunk starts on a strip boundary: lanes 8j .. 8j + 7 are one strip's rows)
        for (int tb = 0; tb < ntc; tb += 32) {                   // wave-uniform
            const int nb = min(ntc - tb, 32);
            const uint32_t colmask = nb >= 32 ? ~0u : ((1u << nb) - 1u);
            uint32_t iv = m_inter, cand = m_and, cor = m_or;
            if (tb > 0) {                                        // further blocks of a wide path: analytic rows exactly, sampled rows as partial
                iv = 0; cand = 0; cor = 0;
                if (mode == ROW_FULL) { uint32_t i2, c2; full_row_masks(tc0 + tb, i2, c2); iv = i2; cand = c2; cor = c2; }
                else if (mode == ROW_SUB) iv = ~0u;
            }
            // tile columns that lie inside the path's rectangle with all their (in-frame) pixel columns
            uint32_t inside = colmask;
            if (tb == 0 && P.x_min > tc0 * TILE_W) inside &= ~1u;
            if (tb + nb == ntc && P.x_max < min((tc1 + 1) * TILE_W, width)) inside &= ~(1u << (nb - 1));
            uint32_t mp = 0, mn = 0, me = 0, mh = 0;
            if (in_frame) {
                if (!in_rows) mn = colmask;
                else if (slow) { mp = mn = me = colmask; }       // not known yet: the general route is always right
                else {
                    iv &= colmask; cand &= colmask; cor &= colmask;
                    mp = iv | (cor & ~cand) | (cand & ~inside);
                    me = iv | cor;
                    mh = colmask & ~me;
                    mn = mp | mh;
                }
            }
            const unsigned long long pb = __ballot(in_rows && (slow || (iv & colmask) != 0u));     // rows with a boundary of the path somewhere in these columns
            // OR over the strip's eight lanes (half a DPP row): mirror the half, then two quad permutations; every lane is active here
#define R3_OR8(v) do { v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false); \
                       v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, false); \
                       v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, false); } while (0)
            R3_OR8(mp); R3_OR8(mn); R3_OR8(me); R3_OR8(mh);
#undef R3_OR8
            mp |= mh & me;                                       // rows without coverage beside rows with: partial even if no single pixel is
            // lanes 8j + t write the class bytes of the strip's tile columns t, t + 8, ...
            for (int t0 = 0; t0 < nb; t0 += 8) {                 // wave-uniform
                const int t = t0 + tsub;
                uint32_t f = (((mp >> t) & 1u) ? CLS_PARTIAL : 0u) | (((mn >> t) & 1u) ? CLS_NOTFULL : 0u) | (((me >> t) & 1u) ? CLS_NONEMPTY : 0u);
                if (f == CLS_NONEMPTY) f |= opq;                 // a full cover that hides what lies below
                const int tc = tc0 + tb + t;
                if (t < nb && band_ok && lane < chunk_rows) {
                    out[(uint32_t)(tc * STRIPS_PER_TILE + strip_in_tile) * n_b] = (uint8_t)f;          // (< 2^32: strips of a tile-row x its entries)
                    if (own_band) {
                        const uint32_t strip_id = (local_trow * (uint32_t)FR->tiles_x + (uint32_t)tc) * STRIPS_PER_TILE + (uint32_t)strip_in_tile;
                        strip_top_note(FR, strip_id, pos1, f, st.pixel);
                        // the tile's strips get heavier by the rows of this path with a boundary (the tile pass starts its heaviest strips first)
                        if (cost_order && (f & CLS_PARTIAL)) {
                            const uint32_t wgt = (uint32_t)__popcll((pb >> (lane & ~7)) & 0xffull);
                            if (wgt) atomicAdd(&FR->strip_cost[strip_id], wgt);
                        }
                    }
                }
            }
        }
    }
    TRACE(7);
    TRACE_OUT(1, block);
}
