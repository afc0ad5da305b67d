#!/usr/bin/env python
"""Stamp profiles/*_traffic.json with the commit they were taken at.

The GPU box has no .git, so tools/summarize_profile.py records only the digest of mmfn_amd/csrc/* the profiled library was built
from.  Run here after copying a round's summaries into profiles/: every traffic file whose digest equals the working tree's gets
`commit` = the commit that last touched mmfn_amd/csrc (plus "+dirty" when csrc has uncommitted changes).

  python tools/stamp_profiles.py [profiles/r05a_traffic.json ...]
"""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from bench import csrc_digest  # noqa: E402


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    now = csrc_digest()
    commit = subprocess.check_output(["git", "-C", ROOT, "log", "-1", "--format=%h", "--", "mmfn_amd/csrc"]).decode().strip()
    if subprocess.call(["git", "-C", ROOT, "diff", "--quiet", "HEAD", "--", "mmfn_amd/csrc"]) != 0:
        commit += "+dirty"
    for f in files:
        rec = json.load(open(f))
        if rec.get("csrc_digest") == now and not rec.get("commit"):
            rec["commit"] = commit
            json.dump(rec, open(f, "w"), indent=1)
            print("%s: commit %s" % (os.path.basename(f), commit))


if __name__ == "__main__":
    main()
