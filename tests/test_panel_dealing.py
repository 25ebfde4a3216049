"""CPU: the panel kernel's dealing of row blocks to waves (csrc/gemmp.hip, row_block_of), read from the source: every table
is a permutation, the four waves (three at 12 blocks) a SIMD holds -- w, w + 4, w + 8, w + 12 -- carry the same number of
k units in BOTH stages (block r: r + 1 in stage 1, nb - r in stage 2), and every SIMD holds the same number of the
fetching waves (upper half of the blocks), which prepare the next panel while the others finish stage 2."""
import os
import re

from conftest import ROOT


def tables():
    src = open(os.path.join(ROOT, "gpzoo_amd", "csrc", "gemmp.hip")).read()
    out = {}
    for nb, body in re.findall(r"NB == (\d+)\) \{ constexpr int t\[\d+\] = \{([0-9, ]+)\}; return t\[w\]; \}", src):
        out[int(nb)] = [int(x) for x in body.split(",")]
    return out


def test_row_block_tables_balance_the_simds():
    t = tables()
    assert set(t) == {12, 16}
    for nb, tab in t.items():
        assert sorted(tab) == list(range(nb))
        s1 = [sum(tab[w] + 1 for w in range(s, nb, 4)) for s in range(4)]
        s2 = [sum(nb - tab[w] for w in range(s, nb, 4)) for s in range(4)]
        fetchers = [sum(1 for w in range(s, nb, 4) if tab[w] >= nb // 2) for s in range(4)]
        if nb == 16:
            assert s1 == [34] * 4 and s2 == [34] * 4 and fetchers == [2] * 4
            # the two longest waves of a SIMD differ by at most 3 units in either stage (the longest ends the stage alone)
            for s in range(4):
                for lens in ([tab[w] + 1 for w in range(s, nb, 4)], [nb - tab[w] for w in range(s, nb, 4)]):
                    top = sorted(lens)[-2:]
                    assert top[1] - top[0] <= 3
        else:
            assert max(s1) - min(s1) <= 1 and max(s2) - min(s2) <= 1 and sum(fetchers) == nb // 2
