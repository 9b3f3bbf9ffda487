"""Host-side model of csrc/panel.hip's index arithmetic (no GPU): the split of K into parts (splits x halves) covers every k-step
exactly once in whole groups and whole k-step pairs; the activation pieces a block's waves request tile the LDS image exactly once, in
the [k-pair][row tile][16 rows][128 B] order the fragment reads assume; the fragment reads return the chunk the MFMA operand needs; the
B fragment ends up in natural k order."""
import itertools


def lds_row_swizzle(row):  # common.hpp
    return (((row >> 1) ^ (row >> 4)) & 3) | (((row ^ (row >> 3)) & 1) << 2)


KTS, NW = 8, 4


def test_k_parts_cover_every_k_step_once_in_whole_groups():
    for K, S, KH, SPG in itertools.product((1024, 2112, 4096, 11008, 28672), (1, 2, 3, 4, 8), (1, 2), (1, 2, 4)):
        T = K // 32
        if (K // 32) % 2 or K % (32 * SPG):
            continue
        align = 2 if SPG < 2 else SPG
        chunk = ((T + S * KH - 1) // (S * KH) + align - 1) // align * align
        tiles = (chunk + KTS - 1) // KTS
        seen = [0] * T
        for part in range(S * KH):
            t0 = part * chunk
            t1 = min(t0 + chunk, T)
            assert t0 % SPG == 0 and t0 % 2 == 0
            for kt in range(tiles):                       # every wave of the block runs the same number of tiles
                tb = t0 + kt * KTS
                assert tb % SPG == 0                      # a tile starts on a group boundary
                for kp in range(KTS // 2):
                    if tb + 2 * kp < t1:                  # a live k-pair: both of its k-steps are inside the matrix and the part
                        assert tb + 2 * kp + 1 < T
                        seen[tb + 2 * kp] += 1
                        seen[tb + 2 * kp + 1] += 1
        assert seen == [1] * T, (K, S, KH, SPG)


def test_activation_pieces_tile_the_image_once():
    for MT in (1, 2, 4, 8):
        KP = KTS // 2
        ppw = KP * MT * 2 // NW
        nmt = MT // 2 if MT >= 2 else 1
        image = {}
        for wave in range(NW):
            for r in range(ppw):
                q = wave + NW * r
                kp = (r * NW // 2 + (wave >> 1)) // MT
                u = (r * NW // 2 // 2) % nmt
                mt = 0 if MT == 1 else ((wave >> 1) + 2 * u) % MT
                h = wave & 1
                assert (q >> 1) == kp * MT + mt and (q & 1) == h, (MT, wave, r)   # the piece lands at q * 1024 = [kp][mt][h]
                for lane in range(64):
                    row = 8 * h + (lane >> 3)
                    logical = (lane & 7) ^ lds_row_swizzle(row)
                    slot = (q * 1024 + lane * 16) // 16
                    assert slot not in image
                    image[slot] = (kp, mt, row, logical)
        assert sorted(image) == list(range(KP * MT * 128))
        # fragment reads: k-step s of the tile, row tile mt, lane (g, i) -> row i, chunk 4 (s & 1) + g of k-pair s >> 1
        for s in range(KTS):
            for mt in range(MT):
                for lane in range(64):
                    g, i = lane >> 4, lane & 15
                    addr = ((s >> 1) * MT + mt) * 2048 + i * 128 + (((4 * (s & 1) + g) ^ lds_row_swizzle(i)) << 4)
                    assert image[addr // 16] == (s >> 1, mt, i, 4 * (s & 1) + g)


def test_b_fragment_is_put_into_natural_k_order():
    # registers after the nibble extraction: b0 = (k0, k4), b1 = (k1, k5), b2 = (k2, k6), b3 = (k3, k7) as (low half, high half)
    b = [(0, 4), (1, 5), (2, 6), (3, 7)]
    perm_lo = lambda hi, lo: (lo[0], hi[0])   # v_perm_b32(hi, lo, 0x05040100): the low halves, lo's first   # noqa: E731
    perm_hi = lambda hi, lo: (lo[1], hi[1])   # v_perm_b32(hi, lo, 0x07060302): the high halves              # noqa: E731
    c0, c2 = perm_lo(b[1], b[0]), perm_hi(b[1], b[0])
    c1, c3 = perm_lo(b[3], b[2]), perm_hi(b[3], b[2])
    assert [k for pair in (c0, c1, c2, c3) for k in pair] == list(range(8))


def test_lds_budget_of_every_built_form():
    for MT, KH in ((1, 2), (2, 2), (4, 2), (8, 1)):
        tile = (KTS // 2) * MT * 2048
        alloc = KH * 2 * tile
        assert alloc <= 160 * 1024
        handoff = NW * MT * 4 * 64 * 4 if KH == 2 else 0          # the second K half's accumulators
        staging = 16 * MT * (64 + 8) * 2                          # the epilogue's [rows][64 + 8] halves
        ticket = MT * 4096
        assert handoff <= ticket and staging <= ticket and ticket + 4 <= alloc, (MT, KH)
        assert tile // NW % 1024 == 0                             # the bf16 conversion pass: whole 1 KB rows per wave
