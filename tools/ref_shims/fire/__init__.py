"""python-fire is not installed in the build container; the reference's scripts only use fire.Fire(main) under
`if __name__ == '__main__'`, which never runs when they are imported for golden generation."""


def Fire(*a, **k):
    raise RuntimeError("fire shim: command-line dispatch is not available")
