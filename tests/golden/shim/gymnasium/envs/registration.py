def register(*args, **kwargs):
    return None
