def info(*a, **k):
    pass


warning = error = debug = info
