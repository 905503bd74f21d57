"""__graft_entry__.smoke(): one small invocation of the hot path on cuda:0, checked against the oracle."""
import numpy as np


def run():
    import __graft_entry__ as ge
    import oracle_lib as ol
    import synth
    pkg = ge.load_package()
    from ssk_amd import capi
    oracle = ol.load()
    ctx = pkg.Context(0)
    rng = np.random.default_rng(2026)
    a, b = synth.synth_kf_pair(rng, 500, 300)
    vals = (1, 1, 370, 60.0)
    got = ctx.data_association(capi.make_kf_view(**a)[0], capi.make_kf_view(**b)[0], capi.DaParams(*vals))
    exp = oracle.data_association(ol.make_kf_view(**a)[0], ol.make_kf_view(**b)[0], ol.DaParams(*vals))
    assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1]) and got[2] == exp[2]
    print("smoke: data association on cuda:0 matches the oracle bit-exactly (%d tracked of %d)" % (got[2], len(got[1])))
    ctx.close()


if __name__ == "__main__":
    run()
