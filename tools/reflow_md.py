#!/usr/bin/env python
"""Reflows the prose of a Markdown file to at most WIDTH columns (tables, code fences, headings and link-only lines are left
alone; bullet items keep their hanging indent).  usage: python tools/reflow_md.py DESIGN.md [width]"""
import re
import sys
import textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 118
lines = open(path).read().split("\n")
out, block, fence = [], [], False
ITEM = re.compile(r"^(\s*)(\* |- |\d+\. |\([a-z]\) )")


def flush():
    global block
    if not block:
        return
    if any(l.lstrip().startswith("|") for l in block) or all(len(l) <= width for l in block):
        out.extend(block)
        block = []
        return
    items, cur = [], []
    for l in block:
        if ITEM.match(l) and cur:
            items.append(cur)
            cur = []
        cur.append(l)
    items.append(cur)
    for it in items:
        m = ITEM.match(it[0])
        first_indent = ""
        hang = ""
        text = " ".join(x.strip() for x in it)
        if m and m.group(2) in ("* ", "- ") or (m and re.match(r"\d+\. ", m.group(2))):
            first_indent = m.group(1)
            hang = m.group(1) + " " * len(m.group(2))
        else:
            lead = re.match(r"^\s*", it[0]).group(0)
            first_indent = hang = lead
        out.extend(textwrap.wrap(text, width=width, initial_indent=first_indent, subsequent_indent=hang, break_long_words=False, break_on_hyphens=False))
    block = []


for l in lines:
    if l.strip().startswith("```"):
        flush()
        fence = not fence
        out.append(l)
        continue
    if fence or l.startswith("#"):
        flush()
        out.append(l)
        continue
    if l.strip() == "":
        flush()
        out.append(l)
        continue
    block.append(l)
flush()
open(path, "w").write("\n".join(out))
