"""Real text for the repeat-rich regime (the reference's own workload is enwik8, Makefile:35-38, which no box here has): source and
documentation files of THIS image, concatenated in a fixed order.  Test infrastructure: the corpus is only used when its sha-256 is
the one on file (tests/golden/realtext.json, made by tests/golden/make_golden_realtext.py in the build container -- same image)."""
import os


def corpus(limit):
    out, total = [], 0
    for root in ("/usr/lib/python3/dist-packages", "/usr/lib/python3.10", "/usr/share/doc", "/usr/include"):
        for d, _, files in sorted(os.walk(root)):
            for f in sorted(files):
                if not f.endswith((".py", ".txt", ".h", ".hpp", ".md", ".rst", ".c", ".json", ".html")):
                    continue
                p = os.path.join(d, f)
                try:
                    b = open(p, "rb").read()
                except OSError:
                    continue
                out.append(b)
                total += len(b)
                if total >= limit:
                    return b"".join(out)[:limit]
    return b"".join(out)
