// NOT COMPILED -- kept for the record (profiles/r05_b3_ring_vs_barrier.txt).  Round 5's barrier-free form of
// gemm_win_b3_stream_kernel<.., DIRW = true>: drop it into rstnet_amd/csrc/gemm_win.hip behind that kernel (it uses its helpers) and
// launch it with 4 * 3 * 128 * B3_RS * 2 + 16 bytes of dynamic LDS to reproduce the measurement.  Results equal the shipped kernel's bit
// for bit (same k order, same product order); it is 5 - 15 % slower on every layer of the codec step.
// ---- round 5: the same stream of k-tiles WITHOUT a workgroup barrier per stage ("ring") ----------------------------------------------
// With the weights out of LDS (DIRW above) the only thing the waves of a workgroup share is the split activation tile, and what the
// barrier of a stage enforces is (a) k-tile g + 1 is completely written before anybody reads it, (b) nobody still reads the buffer
// that is written next.  With two buffers both hold with zero slack: every stall of one wave (its own weight fragments, its own
// activation rows) stops all of them (the ablation table prices the barriers at 12 %).  Here the activation planes live in a ring of
// FOUR buffers: stage g multiplies buffer g % 4 and writes k-tile g + 2 into buffer (g + 2) % 4.  Both conditions become "every wave
// has finished stage g - 2", i.e. one full stage of slack, and are checked against monotonic arrival counters in LDS, one per ring slot
// (a single running sum would not do: waves that are ahead would vouch for one that is behind; a slot's counter can only be ahead
// of NW x its completed stages by waves that passed the very check it backs).  One ds_add per wave and stage; DS instructions of a wave
// execute in order, so the add follows the stage's reads and writes.  The counter is
// sampled late in the previous stage, so the check at the top of a stage normally costs a scalar compare; it spins (re-reading LDS)
// only when a wave is more than a stage behind.  Waves never meet again after the barrier behind the start-up staging: the epilogues
// and the crossings into the next tile drift by up to two stages.  Arithmetic (k order, product order, accumulation) is that of DIRW.
template <bool ELU, bool MASK, int NWN>
__global__ __launch_bounds__(128 * NWN) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_win_b3_ring_kernel(const GemmWinParams p, const int tiles) {
    constexpr int TM = 2, TN = 2;
    constexpr int NT = 128 * NWN, NW = NT / 64;
    constexpr int BM = 128, BN = 64 * NWN;
    constexpr int RA = 512 / NT;
    constexpr int A_PLANE = BM * B3_RS;
    constexpr int BUF = 3 * A_PLANE;               // shorts per ring slot (18 432 bytes)
    constexpr int NBUF = 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    short* const lds = reinterpret_cast<short*>(smem);
    unsigned* const arrive = reinterpret_cast<unsigned*>(lds + NBUF * BUF);       // [4]: arrivals at the end of the stages g with g % 4 == i
    volatile unsigned* const arrive_v = arrive;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / NWN, wn = wave % NWN;
    const int lrow = b3_row(tid >> 2);
    const int lk = (tid & 3) * 4;
    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 8;

    const int M = p.B * p.T_out;
    const int TC = p.T_in * p.C;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int nk = p.K / B3_KB;                    // a multiple of 4 (K % 64 == 0): checked by the launcher

    const int xcd = blockIdx.x & 7;
    const int first = gw_xcd_first(tiles, xcd);
    const int count = gw_xcd_first(tiles, xcd + 1) - first;
    const int stride = ((int)gridDim.x + 7 - xcd) >> 3;
    int l = blockIdx.x >> 3;
    if (l >= count) return;

    struct Ctx {
        unsigned ao[RA];
        int klo[RA], kn[RA];
    };
    auto setup = [&](int m0, Ctx& c) {
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const int m = m0 + lrow + (NT / 4) * j;
            const int mm = min(m, M - 1);
            const int b = mm / p.T_out;
            const int t = mm - b * p.T_out;
            const int f0 = (t * p.S - p.P) * p.C;
            long off = (long)b * p.x_bstride + f0 + lk;
            if (MASK) {
                int klo = max(0, -f0) / B3_KB, khi = min(p.K, TC - f0) / B3_KB;
                if (m >= M || klo >= khi) {
                    klo = khi = 0;
                    off = lk;
                }
                c.klo[j] = klo;
                c.kn[j] = khi - klo;
                off += (long)klo * B3_KB;
            }
            c.ao[j] = (unsigned)(off * 4);
        }
    };
    auto w_tile = [&](int n0) { return (unsigned)(n0 / 32) * (unsigned)nk * 3072u; };
    const unsigned wo = (unsigned)(wn * TN) * (unsigned)nk * 3072u + (unsigned)lane * 16u;
    const unsigned wo_blk = (unsigned)nk * 3072u;

    f32x4 ra[4][RA];
    int rm[4][RA];
    u32x4 rw[2][TN][3];
    const int a_dst = lrow * B3_RS + lk;
    const int a_frag = (wm * TM * 32 + frag_row) * B3_RS + frag_k;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<short*>(p.w3), 0, 0xffffffff, 0x00020000);
    auto load_a = [&](const Ctx& c, const int kt, f32x4 (&dst)[RA], int (&mask)[RA]) {
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            if (MASK) {
                const unsigned d = (unsigned)(kt - c.klo[j]);
                const bool ok = d < (unsigned)c.kn[j];
                mask[j] = ok ? -1 : 0;
                const unsigned vo = c.ao[j] + (ok ? d : 0u) * (B3_KB * 4);
                dst[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, vo, 0, 0));
            } else {
                dst[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, c.ao[j], kt * (B3_KB * 4), 0));
            }
        }
    };
    auto load_wq = [&](auto UU, auto QQ, const unsigned wb, const int kt) {
        constexpr int U = decltype(UU)::value, q = decltype(QQ)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j)
            rw[U][j][q] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wo + j * wo_blk, wb + (unsigned)kt * 3072u + q * 1024u, 0);
    };
    auto take_a = [&](const f32x4 src, const int mask, f32x2 (&v)[2]) {
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        f32x4 x = src;
        if (MASK) x = __builtin_bit_cast(f32x4, __builtin_bit_cast(i32x4, x) & mask);
        if (ELU) {
            x[0] = rst_elu(x[0]); x[1] = rst_elu(x[1]); x[2] = rst_elu(x[2]); x[3] = rst_elu(x[3]);
        }
        v[0] = f32x2{x[0], x[1]};
        v[1] = f32x2{x[2], x[3]};
    };

    f32x16 acc[TM][TN];
    auto clear = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    };

    unsigned seen = 0;                // the arrival counter as last sampled

    // stage S (stream position g, g % 4 == S): ring slot S into the accumulators; register set (S + 2) % 4 -- k-tile g + 2 -- split into
    // slot (S + 2) % 4; the activation rows of stream position g + 5 requested into the set split one stage ago; the weight planes of
    // k-tile g + 2 re-requested into the set this stage multiplies, each after its last product
    auto stage = [&](auto SS, const Ctx& c, const int kta, const unsigned bp, const int ktb, const unsigned need) {
        constexpr int S = decltype(SS)::value;
        constexpr int SP = (S + 2) % 4, F = (S + 1) % 4, SB = S % 2;
        load_a(c, kta, ra[F], rm[F]);
        // every wave has finished stage g - 2 (its reads of the slot written below, its writes of the slot read below)?
        {
            unsigned have = __builtin_amdgcn_readfirstlane(seen);
            while ((int)(have - need) < 0) have = __builtin_amdgcn_readfirstlane(arrive_v[SP]);      // (stage g - 2 ended on counter (g - 2) % 4)
            asm volatile("" ::: "memory");
        }
        bf16x8 fa[TM][3];
        const short* rd = lds + S * BUF;
        short* wr = lds + SP * BUF;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i][2] = *reinterpret_cast<const bf16x8*>(rd + a_frag + 2 * A_PLANE + i * 32 * B3_RS);
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i][1] = *reinterpret_cast<const bf16x8*>(rd + a_frag + A_PLANE + i * 32 * B3_RS);
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i][0] = *reinterpret_cast<const bf16x8*>(rd + a_frag + i * 32 * B3_RS);
        __builtin_amdgcn_sched_barrier(0);

        constexpr int QA[6] = {2, 1, 0, 0, 1, 0};      // weight plane 0's three products first, then plane 2, then plane 1 (see DIRW)
        constexpr int QB[6] = {0, 0, 0, 2, 1, 1};
        f32x2 v[RA][2];
        u32x2 h[RA];
        constexpr int PB = 2 * RA + 1, NOPS = 3 * PB;       // 15 (RA = 2: slots 1 .. 15) or 9 (RA = 1: the odd slots 1 .. 17)
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            const int t = m >> 2, i = (m >> 1) & 1, j = m & 1;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][QA[t]], __builtin_bit_cast(bf16x8, rw[SB][j][QB[t]]), acc[i][j], 0, 0, 0);
            if (m == 11) load_wq(b3_int<SB>{}, b3_int<0>{}, bp, ktb);
            if (m == 15) load_wq(b3_int<SB>{}, b3_int<2>{}, bp, ktb);
            if (m == 23) load_wq(b3_int<SB>{}, b3_int<1>{}, bp, ktb);
            if (m == 0) {
#pragma unroll
                for (int r = 0; r < RA; ++r) take_a(ra[SP][r], rm[SP][r], v[r]);
            }
            const int op = RA == 2 ? m - 1 : ((m & 1) ? (m - 1) >> 1 : -1);
            if (op >= 0 && op < NOPS) {
                const int q = op / PB, w = op % PB;
                if (w < 2 * RA) {
                    h[w >> 1][w & 1] = b3_peel(v[w >> 1][w & 1]);
                } else {
#pragma unroll
                    for (int r = 0; r < RA; ++r) *reinterpret_cast<u32x2*>(wr + a_dst + q * A_PLANE + r * (NT / 4) * B3_RS) = h[r];
                }
            }
            if (m == 19) {          // this wave's reads and writes of the stage are issued: it has arrived (DS operations execute in order)
                asm volatile("" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(arrive + S, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (m == 21) seen = arrive_v[(S + 3) % 4];      // the counter the next stage checks ((g + 1) - 2), compared at its top
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // request cursor of the activation rows: its own walk over the tiles (five stream positions ahead of the products, it can be two
    // tiles ahead when K = 64)
    Ctx c;
    int lr = l;
    int tile = first + l;
    int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    unsigned bp = w_tile(n0), bpn = bp;
    auto cross = [&]() {
        lr += stride;
        if (lr < count) {
            const int tr = first + lr;
            setup((tr / tiles_n) * BM, c);
            bpn = w_tile((tr % tiles_n) * BN);
        }
    };
    setup(m0, c);
    int kr;
    {
        // start of the run: k-tiles 0 and 1 of the first tile into ring slots 0 and 1, the weight fragments of both into sets 0 and 1,
        // activation rows of k-tiles 2, 3 and of stream position 4 into sets 2, 3, 0
        if (tid < 4) arrive[tid] = 0u;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            f32x4 xa[RA];
            int xm[RA];
            load_a(c, g, xa, xm);
            short* wr = lds + g * BUF;
#pragma unroll
            for (int j = 0; j < RA; ++j) {
                f32x2 v[2];
                take_a(xa[j], xm[j], v);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    u32x2 hh;
                    hh[0] = b3_peel(v[0]);
                    hh[1] = b3_peel(v[1]);
                    *reinterpret_cast<u32x2*>(wr + a_dst + q * A_PLANE + j * (NT / 4) * B3_RS) = hh;
                }
            }
        }
        load_wq(b3_int<0>{}, b3_int<0>{}, bp, 0); load_wq(b3_int<0>{}, b3_int<2>{}, bp, 0); load_wq(b3_int<0>{}, b3_int<1>{}, bp, 0);
        load_wq(b3_int<1>{}, b3_int<0>{}, bp, 1); load_wq(b3_int<1>{}, b3_int<2>{}, bp, 1); load_wq(b3_int<1>{}, b3_int<1>{}, bp, 1);
        load_a(c, 2, ra[2], rm[2]);
        load_a(c, 3, ra[3], rm[3]);
        kr = 4;
        if (kr == nk) {
            cross();
            kr = 0;
        }
        load_a(c, kr, ra[0], rm[0]);
        ++kr;
        __syncthreads();
    }
    int kb = 2;
    unsigned nq = 0;                 // NW x the number of whole rotations behind: stage g waits until counter (g - 2) % 4 shows NW arrivals per
                                     // stage of its residue class up to g - 2 (stages 0 and 1 of the run wait for nothing)
    for (;;) {
        const int ln = l + stride;
        const bool has_next = ln < count;
        const int tile_n = first + ln;
        clear();
        for (int kt = 0; kt < nk; kt += 4) {
            stage(b3_int<0>{}, c, kr, bp, kb, nq);
            stage(b3_int<1>{}, c, kr + 1, bp, kb + 1, nq);
            kb += 2;
            if (kb == nk) {
                kb = 0;
                bp = bpn;
            }
            stage(b3_int<2>{}, c, kr + 2, bp, kb, nq + NW);
            kr += 3;
            if (kr == nk) {
                cross();
                kr = 0;
            }
            stage(b3_int<3>{}, c, kr, bp, kb + 1, nq + NW);
            ++kr;
            kb += 2;
            if (kb == nk) {
                kb = 0;
                bp = bpn;
            }
            nq += NW;
        }
        if (!MASK || (m0 + BM <= M && n0 + BN <= p.N)) gw_epilogue<TM, TN, true>(p, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, M, lane);
        else gw_epilogue<TM, TN, false>(p, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, M, lane);
        if (!has_next) break;
        l = ln;
        tile = tile_n;
        m0 = (tile / tiles_n) * BM;
        n0 = (tile % tiles_n) * BN;
    }
}

