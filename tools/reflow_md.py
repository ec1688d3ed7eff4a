#!/usr/bin/env python
"""Re-flow a markdown file's prose to <= WIDTH columns (paragraphs and list items; tables, headings and code blocks untouched)."""
import re
import sys
import textwrap

WIDTH = 118


def main(path):
    lines = open(path).read().split("\n")
    out, para, in_code = [], [], False

    def flush():
        if not para:
            return
        first = para[0]
        m = re.match(r"^(\s*)((?:[*-]|\d+\.)\s+)?", first)
        ind, bullet = m.group(1), m.group(2) or ""
        text = " ".join(l.strip() for l in para)
        text = text[len(bullet):] if bullet and text.startswith(bullet.strip()) else text
        text = text.lstrip("*- ").strip() if bullet and not text else text
        body = " ".join(l.strip() for l in para)
        if bullet:
            body = body[len(bullet.strip()):].strip()
        out.extend(textwrap.wrap(body, width=WIDTH, initial_indent=ind + bullet, subsequent_indent=ind + " " * len(bullet),
                                 break_long_words=False, break_on_hyphens=False))
        para.clear()

    for l in lines:
        if l.startswith("```"):
            flush()
            in_code = not in_code
            out.append(l)
            continue
        if in_code or l.startswith("|") or l.startswith("#") or not l.strip():
            flush()
            out.append(l)
            continue
        if re.match(r"^\s*(?:[*-]|\d+\.)\s+", l):      # a new list item starts a new paragraph
            flush()
        para.append(l)
    flush()
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1])
