"""Transcribe the argument lists of the reference's shipped launch scripts (src/scripts/main_*.sh) into
tests/golden/script_args.json: {script name: [argv tokens]} with the `$sample_idx` loop variable and the `<TODO>`
placeholders replaced by literals.  Data only (flag names and values); runs in the build container."""
import json
import os
import re
import shlex

REF = "/root/reference/src/scripts"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "script_args.json")


def argv_of(path):
    txt = open(path).read()
    m = re.search(r"python main\.py(.*?)(?:\n\s*done|\n\n|\Z)", txt, re.S)
    body = m.group(1).replace("\\\n", " ")
    body = body.replace("$sample_idx", "7").replace("<TODO>/", "/data/").replace("<TODO>", "0")
    body = "\n".join(l for l in body.splitlines() if not l.strip().startswith("#"))
    return shlex.split(body)


def main():
    out = {}
    for f in sorted(os.listdir(REF)):
        if f.endswith(".sh"):
            out[f] = argv_of(os.path.join(REF, f))
    json.dump(out, open(OUT, "w"), indent=1)
    for k, v in out.items():
        print(k, len(v))


if __name__ == "__main__":
    main()
