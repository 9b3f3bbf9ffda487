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


def test_request_counts_fit_the_wait_counter():
    """The hand-counted vmcnt waits: requests per ring slot for every built (strips, bits, group) combination; waits above 63 are
    written as 63 (conservative), and a slot's own requests never exceed the counter."""
    for bits in (4, 3):
        for spg in (2, 4):
            for cpl in (1, 2, 4, 6):
                for zf16 in (False, True):
                    if bits == 3 and cpl == 6:
                        continue
                    z2 = bits == 3 and not zf16
                    lz = cpl * (2 + (1 if z2 else 0))
                    lx = 2 + 2 * cpl * (1 if bits == 4 else 2)
                    l_even, l_odd = lx + (lz if spg == 2 else 0), lx + lz
                    assert max(l_even, l_odd) <= 63
