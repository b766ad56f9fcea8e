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


# what the >= 1 GiB corpus is made of: the four roots of corpus(), the ROCm headers, and forty Python packages by name -- those that the
# build container and the GPU boxes hold byte for byte alike (the container has more: its offline wheelhouse), largest first
BIG_ROOTS = ("/usr/lib/python3/dist-packages", "/usr/lib/python3.10", "/usr/share/doc", "/usr/include", "/opt/rocm/include") + tuple(
    "/usr/local/lib/python3.10/dist-packages/" + pkg for pkg in (
        "torch", "sympy", "plotly", "scipy", "pandas", "triton", "fontTools", "numpy", "sqlalchemy", "matplotlib", "networkx", "kaleido",
        "pygments", "libcst", "textual", "dash_svg", "pymongo", "dash", "mpmath", "narwhals", "pydantic", "hypothesis", "_pytest", "rich",
        "PIL", "torchgen", "dns", "pybind11", "aiohttp", "google", "joblib", "fastapi", "werkzeug", "fsspec", "scikit_build_core",
        "mpl_toolkits", "anyio", "multiprocess", "jinja2", "typer"))


def corpus_big(limit):
    """The same kind of material, from more of the image (ROCm headers, Python packages: torch's sources and headers are most of it), up
    to `limit` bytes: the >= 1 GiB real-text input (tests/golden/realtext_1GiB.json).  Directories in sorted order, depth first; symbolic
    links are not followed."""
    # (the files are read by a pool of threads, a batch of paths at a time and in order: on a box whose page cache is cold the
    # hundred thousand small reads took 90 s one after the other)
    from concurrent.futures import ThreadPoolExecutor

    def paths():
        for root in BIG_ROOTS:
            for d, dirs, files in os.walk(root):
                dirs.sort()
                for f in sorted(files):
                    if f.endswith((".py", ".txt", ".h", ".hpp", ".md", ".rst", ".c", ".json", ".html")):
                        yield os.path.join(d, f)

    def read(p):
        if os.path.islink(p):
            return None
        try:
            return open(p, "rb").read()
        except OSError:
            return None

    out, total, batch = [], 0, []
    with ThreadPoolExecutor(max_workers=32) as ex:
        def drain():
            nonlocal total
            for b in ex.map(read, batch):
                if b is None:
                    continue
                out.append(b)
                total += len(b)
                if total >= limit:
                    return True
            batch.clear()
            return False
        for p in paths():
            batch.append(p)
            if len(batch) >= 4096 and drain():
                return b"".join(out)[:limit]
        if batch and drain():
            return b"".join(out)[:limit]
    return b"".join(out)
