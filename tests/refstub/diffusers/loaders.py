class LoraLoaderMixin:
    pass


class TextualInversionLoaderMixin:
    pass
