"""Build-container script: feeds every search scheme the product library holds — read back through the C-ABI
(cmb_strategy_export_scheme), i.e. the tables the kernels are driven by — to the REFERENCE's own validity checker
(validitychecker/validitychecker.py: Search.covers :131-147, SearchScheme.check_coverage :220-228) and commits its verdicts
as tests/golden/scheme_coverage.json.

The checker is imported from /root/reference where it lies; nothing of it is copied.  The fixture is data: for every
(strategy, k, alternative) the table that was checked (pi, L, U rows), the number of error distributions the reference's
checker enumerated and its verdict.  tests/test_strategy_tables.py::test_scheme_coverage_verdicts then requires, anywhere,
that the library still exports exactly the tables that were judged valid (and re-derives the verdict with an independent
enumeration of its own).

    python tests/golden/make_scheme_verdicts.py
"""
import contextlib
import importlib.util
import io
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/validitychecker/validitychecker.py"

STRATEGIES = ["kuch1", "kuch2", "kianfar", "01*0", "pigeon", "minU", "columba", "multiple_opt"]
MAX_K = 13


def main():
    import columba_amd as ca
    spec = importlib.util.spec_from_file_location("ref_validitychecker", REF)
    vc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(vc)
    out = []
    for name in STRATEGIES:
        st = ca.SearchStrategy(name)
        for k in range(1, MAX_K + 1):
            if not st.supports(k):
                continue
            n_alt, n_parts, _crit = st.describe(k)
            for alt in range(n_alt):
                searches = st.scheme(k, alt)
                with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
                    for pi, lo, up in searches:
                        f.write("{%s} {%s} {%s}\n" % (",".join(map(str, pi)), ",".join(map(str, lo)), ",".join(map(str, up))))
                    path = f.name
                verdict, detail = "valid", ""
                log = io.StringIO()
                try:
                    with contextlib.redirect_stdout(log):
                        vc.SearchScheme(path, k)
                except Exception as e:  # InvalidSearchSchemeError / ValueError of the reference's checker
                    verdict, detail = "invalid", str(e)
                finally:
                    os.unlink(path)
                out.append({"strategy": name, "k": k, "alternative": alt, "parts": n_parts,
                            "searches": [[pi, lo, up] for pi, lo, up in searches],
                            "distributions": vc.n_choose_k(n_parts + k, k), "verdict": verdict, "detail": detail})
                print(name, k, alt, verdict, detail)
    with open(os.path.join(HERE, "scheme_coverage.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    bad = [e for e in out if e["verdict"] != "valid"]
    print(f"{len(out)} schemes checked by the reference's validity checker, {len(bad)} invalid")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
