"""Host-side model of strip_dma.hpp's activation image (no GPU): the LDS-DMA source map and the fragment read map are inverse
permutations (every lane reads the 16-byte chunk its MFMA A fragment needs), and the reads are bank-conflict free under the
lane groups ds_read_b128 is served in (MI355X_MICROARCH.md, LDS section; tools/lab/bank_sim.py uses the same model)."""


def lds_row_swizzle(row):  # common.hpp
    return (((row >> 1) ^ (row >> 4)) & 3) | (((row ^ (row >> 3)) & 1) << 2)


def image_after_dma():
    """byte offset (in 16-byte slots) -> (row, logical 16-byte chunk of the 128-byte row piece) for one stage of one row tile"""
    image = {}
    for h in range(2):               # two pieces of 8 rows x 128 B
        for lane in range(64):
            r = 8 * h + (lane >> 3)
            logical = (lane & 7) ^ lds_row_swizzle(r)     # the chunk this lane FETCHES (swizzle on the source address)
            slot = (h * 1024 + lane * 16) // 16           # LDS-DMA writes lane-linear: base + lane * 16
            assert slot not in image
            image[slot] = (r, logical)
    return image


def test_every_lane_reads_its_fragment_chunk():
    image = image_after_dma()
    assert sorted(image) == list(range(128))              # 2 KB, every slot written once
    for e in range(2):                                    # k-step parity inside the stage
        for lane in range(64):
            g, i = lane >> 4, lane & 15
            addr = i * 128 + (((4 * e + g) ^ lds_row_swizzle(i)) << 4)
            row, chunk = image[addr // 16]
            assert row == i and chunk == 4 * e + g, (e, lane)   # row i, k-slots 8 (4e + g) .. + 7 of the 64-k stage


def test_fragment_reads_are_bank_conflict_free():
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in grp] for grp in groups]
    for e in range(2):
        for grp in groups:
            banks = []
            for lane in grp:
                g, i = lane >> 4, lane & 15
                addr = i * 128 + (((4 * e + g) ^ lds_row_swizzle(i)) << 4)
                banks += [((addr + 4 * d) // 4) % 64 for d in range(4)]
            assert len(set(banks)) == 64, (e, grp)        # 16 lanes x 4 dwords on 64 distinct banks: one LDS cycle


def strip_dma_ring(cpl, spg, bits, zf16, mt=1):  # strip_dma.hpp: strip_dma_ring<CPL, SPG, BITS, ZF16, MT>()
    if spg == 1:
        return 2 if (cpl >= 2 or mt >= 2) else 3
    return 3 if (spg == 2 and (cpl >= 6 or (bits == 3 and cpl >= 4 and not zf16))) else 4


def test_request_counts_fit_the_wait_counter():
    """The hand-counted vmcnt waits (round 5: a ring slot requests its 2 MT activation pieces and ONE word per strip and k-step -- the
    scale / zero words travel in front of the ring, 3-bit fragments are one load): a wait is `vmcnt(all slots - own)`, the counter
    has six bits; whatever is built must fit without the conservative cap of 63 ever being needed."""
    for bits in (4, 3):
        for spg in (1, 2, 4):
            if bits == 3 and spg == 1:
                continue
            for mt, cpls in ((1, (1, 2, 3, 4, 6)), (2, (1,))):
                for cpl in cpls:
                    for zf16 in (False, True):
                        if bits == 3 and cpl == 6 and not (zf16 and spg == 2):
                            continue          # (strip_dma_launch.hpp: six 3-bit strips only with fp16 zero points at 64-wide groups)
                        if spg == 1 and cpl > 2:
                            continue          # (32-wide groups: one or two strips)
                        ns = strip_dma_ring(cpl, spg, bits, zf16, mt)
                        lx = 2 * mt + 2 * cpl
                        assert ns * lx <= 63, (bits, spg, mt, cpl, zf16, ns * lx)
                        assert (2 * ns) % spg == 0     # a round of the ring is whole groups


def test_three_bit_fragment_from_one_load_and_the_lane_below():
    """3 bits: a column's k-step is 96 bits = words 0..2; lane (g, i) owns the 24 bits from bit 24 g.  Each lane loads word (0,1,2,2)[g]
    and takes the lower word of its pair from lane - 16 (g = 1, 2) or itself (g = 0, 3); v_alignbit(own, below, (32 - 8 g) & 31) then
    holds its eight 3-bit values at bits 0, 3, .., 21 (strip_dma.hpp b_frag; round 5: was two loads per fragment)."""
    import random
    rnd = random.Random(5)
    for _ in range(50):
        cols = [[rnd.getrandbits(32) for _ in range(3)] for _ in range(16)]   # column i: its three words
        loaded = {}
        for lane in range(64):
            g, i = lane >> 4, lane & 15
            loaded[lane] = cols[i][2 if g == 3 else g]
        for lane in range(64):
            g, i = lane >> 4, lane & 15
            own = loaded[lane]
            below = loaded[lane - 16 if g in (1, 2) else lane]
            sh = (32 - 8 * g) & 31
            f = (((own << 32) | below) >> sh) & 0xffffffff          # v_alignbit_b32
            stream = cols[i][0] | (cols[i][1] << 32) | (cols[i][2] << 64)
            want = (stream >> (24 * g)) & 0xffffff
            assert f & 0xffffff == want, (lane, hex(f), hex(want))
            for j in range(8):                                       # k = 8 g + j of the k-step
                assert (f >> (3 * j)) & 7 == (stream >> (3 * (8 * g + j))) & 7
