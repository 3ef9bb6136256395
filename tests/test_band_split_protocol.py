"""The band split's progress-word protocol (finc_split.hip, BSP) as a window-level model -- no GPU.

Two workgroups share a problem: workgroup g owns the bands g, g + 2, ... and chains them; band k needs the last rows of band
k - 1, which the OTHER workgroup solves.  The kernel's rules, restated (all in units of windows of four steps, local to a
workgroup; GR = W / 4 store groups per row):
  * at step 0 of its window w a B wave requests piece w + 1 of the rows above (group gq = (w + 1) % GR of its local band
    i = (w + 1) // GR) and, if that band has a band above it in the image, first WAITS until the producer's progress word is at
    least  i' * GR + gq + 6  (i' = the producer's local index of the band above): the group leaves the producer at step 0 of
    its window i' * GR + gq + 5 (rows P-2, P-1: fs4 = -5; the handed-over rows are stored early) and is complete by its step 3;
  * at step 3 of window w it publishes w + 1 (the handed-over stores of the windows 0 .. w are complete);
  * after its last window it drains its stores and publishes "all complete".
A workgroup blocked in a wait publishes nothing further, so the two can wait for each other: band k + 2 starts GR windows after
band k on the same workgroup, and by then band k + 1 must have delivered.  The model runs both workgroups to a fixpoint under
the most favourable scheduling; if it stalls, no timing can save the kernel.  It must complete exactly for the maps the
library sends to the band split (W >= 64: finc_split.hip bsp_nwg) -- and it reproduces the two failures met on the way: the
first version's dead-lock on 32-column maps, and the tail the stress test found at W = 68 when the final publish was missing."""
import pytest

NWG = 2
ALL_DONE = 0xFFFFF


def run_model(H, W, final_publish=True, lookahead=1):
    """Returns True if both workgroups finish.  `lookahead`: piece w + lookahead is requested in window w."""
    P = 16
    NB = (H + P - 1) // P
    GR = W // 4
    NBL = (NB + NWG - 1) // NWG
    T = NBL * W + P - 1
    Tr = (T + 2 + 7) // 8 * 8 - 2                    # finc_split_launch: the B waves' loop is unrolled by 8
    nwin = (Tr + 2) // 4                             # windows 0 .. nwin - 1 (u = 0 .. Tr + 1)
    window = [0, 0]                                  # next window each workgroup will enter
    published = [0, 0]
    done = [False, False]

    def need_for(wg, w):
        piece = w + lookahead
        i, gq = piece // GR, piece % GR
        kb = wg + i * NWG                            # the image band whose rows above are requested
        if kb < 1 or kb >= NB:
            return None
        return ((kb - 1) // NWG) * GR + gq + 6

    def prologue_needs(wg):                          # pieces 0 .. lookahead - 1 are requested before the loop
        out = []
        for piece in range(lookahead):
            i, gq = piece // GR, piece % GR
            kb = wg + i * NWG
            if 1 <= kb < NB:
                out.append(((kb - 1) // NWG) * GR + gq + 6)
        return out

    started = [False, False]
    progress = True
    while progress and not all(done):
        progress = False
        for wg in (0, 1):
            other = (wg + NWG - 1) % NWG
            if done[wg]:
                continue
            if not started[wg]:
                if all(published[other] >= n for n in prologue_needs(wg)):
                    started[wg] = True
                    progress = True
                continue
            w = window[wg]
            if w >= nwin:
                if final_publish:
                    published[wg] = ALL_DONE
                done[wg] = True
                progress = True
                continue
            need = need_for(wg, w)
            if need is not None and published[other] < need:
                continue                             # blocked in progress_wait at step 0 of window w
            published[wg] = max(published[wg], w + 1)   # step 3 of window w
            window[wg] = w + 1
            progress = True
    return all(done)


@pytest.mark.parametrize("W", [64, 68, 72, 96, 128])
@pytest.mark.parametrize("H", [17, 26, 32, 33, 48, 64, 85, 100, 128, 250])
def test_the_protocol_completes_on_every_map_the_library_splits(H, W):
    assert run_model(H, W)


@pytest.mark.parametrize("W", [32, 40, 48])
def test_narrow_maps_dead_lock_which_is_why_the_library_does_not_split_them(W):
    """band k + 2 starts W/4 windows after band k; the producer of band k + 1 is 7 windows behind its consumer's requests and
    needs band k's groups up to 5 windows later: W/4 - 1 >= 12, i.e. W >= 52."""
    assert not run_model(64, W)


def test_the_tail_the_stress_test_found():
    """W = 68: 17 store groups per row; the loop of a one-band producer (H = 26: two bands, one each) ends at window 21 and has
    published 21 (that version said a window's stores complete one window later than today's), the consumer's last piece
    needs 22.  Without the publish behind the loop the consumer waits forever.  (Today's early store of the handed-over rows
    gives the tail one window of slack; the final publish stays -- a longer tail, e.g. W = 76, would need it again.)"""
    assert run_model(26, 68, final_publish=True)
    assert run_model(26, 64, final_publish=False)


def test_the_library_rule_matches_the_model():
    """finc_split.hip bsp_nwg: W >= 64 (host-only query)."""
    from fincflow_amd import _lib
    for W in (32, 48, 60, 64, 68, 72):
        v = _lib.inverse_variant(8, 4, 24, 64, W, 3, 3)
        assert v is not None and v["sec"] == 4
        split = v["workgroups"] == 2 * 8 * 4
        assert split == (W >= 64) and (not split or run_model(64, W)), (W, v)
