"""tools/ladder_model.py [workload] [MiB] -- CPU model of the prefilter as the kernel evaluates it (development only):
how many positions pass level 1, how many of those the prefix ladder is asked about, how many are walked, against the
number of positions that really match.  Uses the tables the library compiled (PFACX_getTable)."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo")
from pfac_amd import api, workloads as wl          # noqa: E402

from tests.filter_model import prefilter_model as filter_model       # noqa: E402


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "c3"
    mib = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    cfg = wl.make_config(name)
    pf = wl.write_pattern_file("/tmp/ladder_model_%s.pat" % name, cfg.patterns)
    h = api.PFAC.createHostOnly()
    h.readPatternFromFile(pf)
    info = h.info()
    data = cfg.input_slice(mib << 20, 0)
    res = h.match_host_array(data)
    l1, cand, walk = filter_model(h, data)
    hit = res != 0
    assert np.all(walk[hit]), "false negative"
    print("%s: %d MiB, gram3 2^%d bits, ladder 2^%d bits (%d set), stops %d go-ons %d thin %d extend %d last %d, tail entries %d (LDS) / %d (device memory, 2^%d buckets), skip tags %d" % (
          name, mib, info.filterLog2Bits, info.filterLog2BitsLadder, info.filterBitsSetLadder, info.ladderStops, info.ladderGoOns, info.ladderThin, info.ladderExtend,
          info.filterLadderLast, info.filterTailEntries, info.filterTailGlobalEntries, info.filterLog2TailGlobal, info.filterSkipTags))
    _, _, walk0 = filter_model(h, data, veto=False)
    print("  without the veto: walks %.5f of positions, per GiB %.2f M" % (walk0.mean(), walk0.mean() * 1073.74))
    print("  level-1 hits %.4f  ladder candidates %.4f  walks %.5f  matches %.5f of positions  -> walks per match %.1f, per GiB %.2f M" % (
        l1.mean(), cand.mean(), walk.mean(), hit.mean(), walk.sum() / max(1, hit.sum()), walk.mean() * 1073.74))
    h.destroy()
