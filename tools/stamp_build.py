"""Evidence hygiene: every JSON / JSONL file of a round's evidence set carries the ids of the kernel sources and of the library it was
measured on (megahit_amd/buildid.py), and one call checks that they all agree.

    python tools/stamp_build.py stamp FILE...     add "build_id" / "lib_id" (top level of a JSON document, every line of a .jsonl)
    python tools/stamp_build.py check FILE...     exit 1 unless every file carries the ids of the CURRENT tree and library"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from megahit_amd.buildid import build_id, lib_id  # noqa: E402


def docs(path):
    with open(path) as f:
        text = f.read()
    if path.endswith(".jsonl"):
        return [json.loads(l) for l in text.splitlines() if l.strip().startswith("{")], True
    return [json.loads(text)], False


def main():
    mode, files = sys.argv[1], sys.argv[2:]
    b, l = build_id(), lib_id()
    bad = 0
    for path in files:
        try:
            ds, lines = docs(path)
        except Exception as ex:
            print("%s: not JSON (%s)" % (path, ex))
            bad += 1
            continue
        if mode == "stamp":
            for d in ds:
                d.setdefault("build_id", b)
                d.setdefault("lib_id", l)
            with open(path, "w") as f:
                if lines:
                    f.write("".join(json.dumps(d) + "\n" for d in ds))
                else:
                    json.dump(ds[0], f, indent=1)
                    f.write("\n")
        else:
            for d in ds:
                if d.get("build_id") != b or d.get("lib_id") not in (l, None):
                    print("%s: build_id %s / lib_id %s, the tree is %s / %s" % (path, d.get("build_id"), d.get("lib_id"), b, l))
                    bad += 1
                    break
    if mode == "check":
        print("%d files, %d with other ids than %s / %s" % (len(files), bad, b, l))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
