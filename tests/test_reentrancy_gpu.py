"""GPU: the C-ABI is re-entrant -- "one HIP stream per host thread" (include/fastga_amd.h; what the
reference, with its file-static tables and thread team, cannot do: RSDsort.c:26-33).  Two host threads, each with a
session of its own on cuda:0, run DIFFERENT comparisons at the same time, several times over; every result must be the
file the same comparison gives when it runs alone.  (ctypes releases the GIL for the duration of a C call, so the two
fga_session_run calls really overlap; the device pool, the streams and the kernels' scratch are what is being shared.)"""
import os
import threading

import pytest

pytestmark = pytest.mark.gpu


def _view(path):
    from oracle import harness as H
    return H.oneview(path)


def test_two_sessions_on_two_host_threads_give_their_serial_results(toy_pair, family_pair, tmp_path):
    from fastga_amd import device as D
    w = str(tmp_path)
    _, ta, tb = toy_pair
    _, fa, fb = family_pair
    jobs = {"toy": (ta, tb, {}), "family": (fa, fb, dict(freq=30)), "toy_self": (ta, None, {}),
            "family_sym": (fa, fb, dict(symmetric=True, freq=6))}
    serial = {}
    for name, (a, b, kw) in jobs.items():                 # each comparison alone
        out = os.path.join(w, name + ".serial.1aln")
        st = D.run(a, b, out, nthreads=4, reference_threads=4, **kw)
        serial[name] = (_view(out), st["nseeds"], st["nalns"], st["nlive"])
        assert st["nlive"] > 0

    errors, results = [], {}

    def worker(names, tag):
        try:
            for rep in range(3):
                for name in names:
                    a, b, kw = jobs[name]
                    ses = D.Session(a, b, nthreads=4)
                    out = os.path.join(w, f"{name}.{tag}.{rep}.1aln")
                    for _ in range(2):                     # the session's second run reuses its buffers
                        st = ses.run(out_path=out, nthreads=4, reference_threads=4, **kw)
                    ses.close()
                    results[(tag, rep, name)] = (_view(out), st["nseeds"], st["nalns"], st["nlive"])
        except Exception as e:                            # noqa: BLE001 -- reported by the main thread
            errors.append((tag, repr(e)))

    t1 = threading.Thread(target=worker, args=(["toy", "family_sym", "toy_self"], "t1"))
    t2 = threading.Thread(target=worker, args=(["family", "toy_self", "toy", "family_sym"], "t2"))
    t1.start(); t2.start()
    t1.join(); t2.join()
    assert not errors, errors
    assert len(results) == 3 * 3 + 3 * 4
    for (tag, rep, name), got in results.items():
        assert got[1:] == serial[name][1:], (tag, rep, name, got[1:], serial[name][1:])
        assert got[0] == serial[name][0], (tag, rep, name)
