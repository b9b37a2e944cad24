class AddOne:
    def __call__(self, x):
        return x + 1


class Double:
    def __call__(self, x):
        return x * 2
