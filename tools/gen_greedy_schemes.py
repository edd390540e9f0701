#!/usr/bin/env python3
"""Prints the C++ rows of host_schemes.hpp: greedy(k) from the reference's data files (tests/golden/search_schemes is a verbatim
copy of /root/reference/search_schemes): pigeon_adapted/<k>/searches.txt for k = 8 .. 13 hold ColumbaSearchStrategy's greedy schemes
(searchstrategy.h:3417-3658).  Digits beyond 9 are written a, b, c, d."""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dig = "0123456789abcdefghijklmnopqrstuvwxyz"
for k in range(8, 14):
    rows = []
    for l in open(os.path.join(ROOT, "tests", "golden", "search_schemes", "pigeon_adapted", str(k), "searches.txt")).read().splitlines():
        if l.strip():
            g = [[int(x) for x in t.strip("{}").split(",")] for t in l.split()]
            rows.append('"' + " ".join("".join(dig[v] for v in t) for t in g) + '"')
    print(f"    case {k}:\n        return schemeFromRows({k}, {{" + ", ".join(rows) + "});")
