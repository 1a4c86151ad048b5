#!/usr/bin/env python3
"""One-off assembler of DESIGN.md for round 6 (kept so that the split is reproducible): preamble + the unchanged sections 1-3, 5, 6, 8, 9 of the
round-5 file (read from /tmp/design_keep.json, written by the extraction step in the session log) + the sections written new below.
The history that used to live in sections 4, 7, 9a-c and 10 is DESIGN_HISTORY.md."""
import json
import sys

keep = json.load(open("/tmp/design_keep.json"))
new = open(sys.argv[1]).read()        # the hand-written new sections with markers <<SEC123>> <<DIST>> <<SEC56>> <<SEC8>> <<SEC9>>
for k, m in (("sec123", "<<SEC123>>"), ("dist", "<<DIST>>"), ("sec56", "<<SEC56>>"), ("sec8", "<<SEC8>>"), ("sec9", "<<SEC9>>")):
    assert m in new, m
    new = new.replace(m, keep[k])
open("DESIGN.md", "w").write(new)
print("wrote DESIGN.md:", new.count("\n"), "lines")
