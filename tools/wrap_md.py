#!/usr/bin/env python
"""Re-wrap the paragraphs and list items of a Markdown file to a column limit (tables, headings and blank lines stay as they are).
    python tools/wrap_md.py DESIGN.md [118]"""
import re
import sys
import textwrap


def main(path, width=118):
    lines = open(path, encoding="utf8").read().split("\n")
    res, para = [], []

    def flush():
        if not para:
            return
        first = para[0]
        m = re.match(r"^(\s*(?:[*-]|\d+\.)\s+)", first)
        ind = m.group(1) if m else ""
        text = " ".join([first[len(ind):]] + [ln.strip() for ln in para[1:]])
        res.extend(textwrap.wrap(text, width=width, initial_indent=ind, subsequent_indent=" " * len(ind), break_long_words=False, break_on_hyphens=False))
        del para[:]
    fence = False
    for ln in lines:
        if ln.startswith("```"):
            flush()
            fence = not fence
            res.append(ln)
            continue
        if fence or ln.startswith("|") or ln.startswith("#") or ln.strip() == "":
            flush()
            res.append(ln)
            continue
        if re.match(r"^\s*(?:[*-]|\d+\.)\s+", ln):
            flush()
        para.append(ln)
    flush()
    open(path, "w", encoding="utf8").write("\n".join(res))
    print(path, "lines over the limit:", [(i + 1, len(l)) for i, l in enumerate(res) if len(l) > width + 2])


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 118)
