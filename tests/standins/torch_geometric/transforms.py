class NormalizeFeatures:
    def __call__(self, data):
        return data
