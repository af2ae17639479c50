"""Stand-in for termcolor (see ../README.md)."""


def colored(text, *a, **k):
    return text
