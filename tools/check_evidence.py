#!/usr/bin/env python3
"""One hash over the sources a profile set depends on (ideas_amd/csrc/*, ideas_amd/**/*.py, bench.py, include/*.h).
    python tools/check_evidence.py --write <file>     (tools/collect_profiles.sh, on the GPU box, before the measurements)
    python tools/check_evidence.py <file>             (before committing profiles/: exit 1 if the tree differs from what was measured)"""
import glob
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tree_hash():
    files = sorted(glob.glob(os.path.join(ROOT, "ideas_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "ideas_amd", "csrc", "*.hpp")) +
                   glob.glob(os.path.join(ROOT, "ideas_amd", "**", "*.py"), recursive=True) + glob.glob(os.path.join(ROOT, "include", "*.h")) +
                   [os.path.join(ROOT, "bench.py")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16], len(files)


if __name__ == "__main__":
    sha, n = tree_hash()
    if sys.argv[1] == "--write":
        open(sys.argv[2], "w").write("%s %d files\n" % (sha, n))
        print("evidence source hash", sha, n, "files")
    else:
        want = open(sys.argv[1]).read().split()[0]
        if want != sha:
            print("STALE: the profile set was measured at source hash %s, the tree is at %s" % (want, sha))
            sys.exit(1)
        print("profile set matches the tree (%s)" % sha)
