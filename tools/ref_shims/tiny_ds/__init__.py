"""A three-line dataset both the reference's and this package's build() can construct (golden generation and the
ConcatDataset test)."""


class Tiny:
    def __init__(self, n, tag, scale=1):
        self.n, self.tag, self.scale = n, tag, scale

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return dict(tag=self.tag, value=i * self.scale)
