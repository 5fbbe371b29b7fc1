def valmap(f, d):
    return {k: f(v) for k, v in d.items()}


def keymap(f, d):
    return {f(k): v for k, v in d.items()}
