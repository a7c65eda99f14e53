"""Text metric of the reference's validation loop (`desta/utils/metrics.py:3-32`, used by `desta_trainer.py:36, 205-209`).

`ConsecutiveWordsAccuracyMetric()(pred, label)` is True when the normalised label occurs in the normalised prediction as a run
of consecutive words.  The reference normalises with `whisper_normalizer.basic.BasicTextNormalizer` (third-party, absent
here): that package republishes OpenAI Whisper's basic normaliser, whose algorithm is restated below and pinned in
tests/test_metrics.py against the copy of the same normaliser that `transformers` ships.
"""
import re
import unicodedata


def remove_symbols(s: str) -> str:
    """NFKC, then every mark / symbol / punctuation code point (Unicode categories M*, S*, P*) becomes a space."""
    return "".join(" " if unicodedata.category(c)[0] in "MSP" else c for c in unicodedata.normalize("NFKC", s))


class BasicTextNormalizer:
    """lower-case, drop `<...>` / `[...]` spans and parenthesised spans, symbols to spaces, whitespace runs to one space."""

    def __call__(self, s: str) -> str:
        s = s.lower()
        s = re.sub(r"[<\[][^>\]]*[>\]]", "", s)
        s = re.sub(r"\(([^)]+?)\)", "", s)
        s = remove_symbols(s).lower()
        return re.sub(r"\s+", " ", s)


class ConsecutiveWordsAccuracyMetric:
    metric_name = "consecutive_words_accuracy"

    def __init__(self):
        self.normalizer = BasicTextNormalizer()

    def __call__(self, pred: str, label: str) -> bool:
        return self.check_consecutive_words(long_string=self.normalizer(pred), short_string=self.normalizer(label))

    @staticmethod
    def check_consecutive_words(long_string: str, short_string: str) -> bool:
        hay, needle = long_string.lower().split(), short_string.lower().split()
        n = len(needle)
        return any(hay[i:i + n] == needle for i in range(len(hay) - n + 1))     # an empty label matches any prediction
