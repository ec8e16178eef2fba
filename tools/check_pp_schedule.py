#!/usr/bin/env python
"""Static check of gemm_pp.hip's DMA / read / wait schedule (no GPU): replays the phase program of both tile shapes for a
stream of K-tiles and asserts, for the two wave groups running one barrier apart,
  RAW: every wave has waited (counted vmcnt) for its pieces of a block at least one barrier before ANY wave reads it,
  WAR: a block is issued into an LDS region only >= 2 phases after the last read of the block it overwrites,
  vmcnt: the count used at a wait really covers the blocks that are needed next (loads retire in order).
Intervals: group G runs the load segment of phase g in interval 2g+G and its MFMA segment in 2g+G+1."""
import itertools


def check(name, phases_per_kt, prologue, program, slot_of, ring_kt, KT=7, tiles=3, extra=None, stores_per_tile=0, issue_in_mma=False):
    """extra (round 5): extra(i, p, tile) -> (n_loads, wait_override, needs_landed) for phase p of the i-th K-tile of tile `tile`
    of the workgroup's walk: n_loads ordinary loads issued BEHIND the phase's DMA block (they sit in the same
    in-order vmcnt queue), an optional vmcnt value replacing the program's, and the (tile, i) whose loads this phase consumes
    (they must have been retired by a wait of an EARLIER phase of the same wave).  stores_per_tile: the epilogue's stores,
    which queue behind the DMA pieces issued before them (gfx9 has no vscnt).
    issue_in_mma (the round-5 experiment that left the kernels in round 6; kept in the checker): a phase's block is issued inside its MFMA burst, i.e. BEHIND the
    phase's wait (and behind the extra loads of its load segment) instead of in front of it; the WAR rule is kept as for an issue
    in the load segment (the real issue is half a phase later: conservative)."""
    """program[p] = (reads, issue, wait) for phase p of K-tile J:
         reads : list of block kinds read from K-tile J in this phase
         issue : (dJ, kind, pieces_per_wave) or None -- block of K-tile J+dJ issued in this phase
         wait  : vmcnt value or None
       prologue = ([(J, kind, pieces)], vmcnt)"""
    total_kt = KT * tiles
    issued = []            # per-wave issue order: (J, kind, pieces, phase_issued)
    waited_until = {}      # (J, kind) -> phase g in whose load segment the wait that covers it sits (-1 = prologue)
    last_read = {}         # (J, kind) -> phase of the read
    issue_phase = {}
    for (J, kind, n) in prologue[0]:
        issued.append((J, kind, n, -1))
        issue_phase[(J, kind)] = -1

    def apply_wait(cnt, g):
        # loads retire in order: everything except the newest `cnt` pieces is complete
        left = cnt
        done_upto = len(issued)
        for i in range(len(issued) - 1, -1, -1):
            if left - issued[i][2] < 0:
                if left > 0:
                    done_upto = i  # some pieces of this block may still be in flight: the block is NOT complete
                break
            left -= issued[i][2]
            done_upto = i
        for (J, kind, n, _) in issued[:done_upto]:
            waited_until.setdefault((J, kind), g)

    apply_wait(prologue[1], -1)
    g = 0
    for J in range(total_kt):
        if extra is not None and stores_per_tile and J > 0 and J % KT == 0:
            issued.append((("S", J // KT), "S", stores_per_tile, g))  # the finished tile's stores
        for p in range(phases_per_kt):
            reads, issue, wait = program[p]
            n_extra, wait_override, needs = extra(J % KT, p, J // KT) if extra is not None else (0, None, None)
            if wait_override is not None:
                wait = wait_override
            if needs is not None:
                key = (("R", J // KT, needs), "R")
                assert key in waited_until and waited_until[key] < g, f"{name}: residual loads {key} consumed in phase {g} before a wait retired them"
            for kind in reads:
                key = (J, kind)
                assert key in waited_until, f"{name}: phase {g} reads {key} which no wait covers"
                # RAW across groups: wait in load segment g_w (interval 2 g_w + G), read in interval 2 g + G'
                assert waited_until[key] <= g - 1, f"{name}: {key} waited in phase {waited_until[key]}, read in phase {g}"
                last_read[key] = g
            if issue_in_mma:  # load segment first: extra loads, then the wait; the block follows inside the MFMAs
                if n_extra:
                    issued.append((("R", J // KT, J % KT), "R", n_extra, g))
                    n_extra = 0
                if wait is not None:
                    apply_wait(wait, g)
                    wait = None
            if issue is not None:
                dJ, kind, n = issue
                Ji = min(J + dJ, total_kt - 1)  # the cursor saturates at the end of the stream (duplicate blocks)
                Jslot = J + dJ
                # WAR: the region of (Jslot, kind) was last used by (Jslot - ring_kt, kind')
                for (Jo, ko), gr in list(last_read.items()):
                    if Jo == Jslot - ring_kt and slot_of(ko) & slot_of(kind):
                        assert g >= gr + 2, f"{name}: block {(Jslot, kind)} issued in phase {g} overwrites {(Jo, ko)} read in phase {gr}"
                # everything of K-tile Jslot-ring that overlaps must already have been read
                for ko in [k for k in all_kinds if slot_of(k) & slot_of(kind)]:
                    if Jslot - ring_kt >= 0 and Jslot - ring_kt < total_kt:
                        assert (Jslot - ring_kt, ko) in last_read, f"{name}: {(Jslot, kind)} overwrites unread {(Jslot - ring_kt, ko)} at phase {g}"
                issued.append((Jslot if Jslot < total_kt else -Jslot, kind, n, g))
                issue_phase[(Jslot, kind)] = g
            if n_extra:
                issued.append((("R", J // KT, J % KT), "R", n_extra, g))
            if wait is not None:
                apply_wait(wait, g)
            g += 1
    cover = [last_read[k] - issue_phase[k] for k in last_read if k in issue_phase and issue_phase[k] >= 0]
    print(f"{name}: OK  ({g} phases; a block is requested {min(cover)}..{max(cover)} phases before it is read)")


# ---- 256 x 256: kinds A0 B0 B1 A1, 2 pieces per wave each, separate LDS regions per kind (x K-tile parity)
all_kinds = ["A0", "B0", "B1", "A1"]
check("pp256", 4,
      ([(0, "A0", 2), (0, "B0", 2), (0, "B1", 2), (0, "A1", 2), (1, "A0", 2), (1, "B0", 2)], 8),
      [(["A0", "B0"], (1, "B1", 2), 8), (["B1"], (1, "A1", 2), 8), (["A1"], (2, "A0", 2), 8), ([], (2, "B0", 2), 8)],
      slot_of=lambda k: {"A0": 1, "B0": 2, "B1": 4, "A1": 8}[k], ring_kt=2)

# ---- 256 x 192: issue blocks I0 (A pieces 0..23), I1 (A 24..31 + B0), I2 (B1 + B2); reads: phase 0 A (= I0 + I1) and B0 (I1),
# phase 1 B1 (I2), phase 2 B2 (I2)
all_kinds = ["I0", "I1", "I2"]
check("pp192", 3,
      ([(0, "I0", 3), (0, "I1", 2), (0, "I2", 2), (1, "I0", 3)], 5),
      [(["I0", "I1"], (1, "I1", 2), 5), (["I2"], (1, "I2", 2), None), (["I2"], (2, "I0", 3), 5)],
      slot_of=lambda k: {"I0": 1, "I1": 2, "I2": 4}[k], ring_kt=2)

# ---- conv_pp128 (conv_pp.hip): 2 phases per K-tile, ring of 3 K-tiles, both blocks of K-tile J+2 requested during K-tile J
all_kinds = ["I0", "I1"]
check("conv_pp128", 2,
      ([(0, "I0", 3), (0, "I1", 3), (1, "I0", 3), (1, "I1", 3)], 6),
      [(["I0", "I1"], (2, "I0", 3), None), (["I1"], (2, "I1", 3), 6)],
      slot_of=lambda k: {"I0": 1, "I1": 2}[k], ring_kt=3)

# ---- 256 x 192 with the residual riding on a tile's first eight K-tiles (gemm_pp192_kernel, EPI bit 2): K-tile i < 6 requests the
# four line pieces of sub-tile i in phase 1 (behind I2) and its phase-2 wait is vmcnt(9); K-tile i + 1 consumes them in phase 1.
def _resid(i, p, tile):
    n = 4 if (p == 1 and i <= 5) else 0
    w = 9 if (p == 2 and i <= 5) else None
    needs = (i - 1) if (p == 1 and 1 <= i <= 6) else None
    return n, w, needs


all_kinds = ["I0", "I1", "I2"]
check("pp192_resid", 3,
      ([(0, "I0", 3), (0, "I1", 2), (0, "I2", 2), (1, "I0", 3)], 5),
      [(["I0", "I1"], (1, "I1", 2), 5), (["I2"], (1, "I2", 2), None), (["I2"], (2, "I0", 3), 5)],
      slot_of=lambda k: {"I0": 1, "I1": 2, "I2": 4}[k], ring_kt=2, KT=12, tiles=3, extra=_resid, stores_per_tile=24)
