"""Counted-vmcnt constants of k_gemm_ph: per-thread loads outstanding when the loads a later phase reads must have landed.
Groups staged per phase position: 3 loads at positions 2 and 3 (A half + B block), 1 elsewhere; group g of tile u is issued at
phase index (u-2)*NJ + g + 2 and first read in phase (u, g) (A, i.e. groups 0 and 1, in phase (u, 0))."""


def sim(NJ, ahead):
    cnt = lambda pos: 3 if pos in (2, 3) else 1
    out = {}
    for j in range(NJ):                                   # the wait sits in the read section of phase position j
        P = 10 * NJ + j
        u, g = divmod(P + ahead, NJ)
        need = [(u, g)] if g >= 1 else [(u, 0), (u, 1)]
        last = max((uu - 2) * NJ + gg + 2 for uu, gg in need)
        assert last <= P
        out[j] = sum(cnt(q % NJ) for q in range(last + 1, P + 1))
    return out


if __name__ == "__main__":
    for NJ in (4, 5):
        print("NJ", NJ, "two barriers per phase (wait for phase j+1):", sim(NJ, 1), "| one barrier (wait for phase j+2):", sim(NJ, 2))
