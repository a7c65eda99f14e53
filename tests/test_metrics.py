"""CPU: the validation text metric (`desta/utils/metrics.py` of the reference).  `whisper_normalizer` is not installed, so the
normaliser is pinned against the copy of the same OpenAI-Whisper basic normaliser that `transformers` ships."""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "desta2.5-audio_amd"))

CORPUS = ["Hello, World! It's <b>bold</b> (ok) [x] café", "The  answer is: B.", "naïve — résumé", "", "   ", "(all in parens)",
          "Ünïcödé ½ ① ﬁ ligature", "tabs\tand\nnewlines", "emoji 😀 + math ∑ − ×", "a[b]c<d>e(f)g", "nested ((x)) [y [z]]",
          "Angry, shouting!!", "The speaker is a woman in her 30s.", "don't can't it's"]


def test_normalizer_matches_the_openai_basic_normalizer():
    from transformers.models.whisper.english_normalizer import BasicTextNormalizer as Ref
    from desta.utils.metrics import BasicTextNormalizer
    ref, mine = Ref(), BasicTextNormalizer()
    for s in CORPUS:
        assert mine(s) == ref(s), s


def test_consecutive_words_accuracy():
    from desta.utils.metrics import ConsecutiveWordsAccuracyMetric
    m = ConsecutiveWordsAccuracyMetric()
    assert m.metric_name == "consecutive_words_accuracy"
    assert m("The answer is: Angry.", "angry") and m("It is (certainly) a DOG barking!", "a dog") and m("x", "")
    assert not m("a dog is here", "dog a") and not m("sad", "very sad") and not m("", "word")
    assert m("The emotion is <neutral> happy", "Happy!") and not m("unhappy", "happy")          # whole words, not substrings


def test_save_results_report(tmp_path):
    """`_save_results` (desta_trainer.py:191-251): preds JSONL + report JSON, per-sample and per-category accuracy."""
    from desta.trainer.desta_trainer import DeSTA25Trainer
    from desta.utils.metrics import ConsecutiveWordsAccuracyMetric
    me = types.SimpleNamespace(metrics=ConsecutiveWordsAccuracyMetric(), cfg={"exp_dir": str(tmp_path), "name": "x"})
    results = [dict(prediction="it is a cat", label="cat", category="animal", context="c1", audio_context="a1"),
               dict(prediction="a dog", label="cat", category="animal", context="c2"),
               dict(prediction="angry voice", label="Angry", category="emotion", context="c3"),
               dict(prediction="no category", label="category", context="c4")]
    path = tmp_path / "results" / "val" / "val@ep=1.0-3.jsonl"
    rep = DeSTA25Trainer._save_results(me, results, path, ckpt="ep=1.0-3")
    assert rep["metric"] == "consecutive_words_accuracy" and rep["accuracy_by_sample"] == 0.75
    assert rep["categories_accuracy"] == {"animal": 0.5, "emotion": 1.0, "all": 1.0}
    assert abs(rep["avg_accuracy_by_category"] - (0.5 + 1.0 + 1.0) / 3) < 1e-12 and rep["ckpt"] == "ep=1.0-3" and rep["name"] == "DeSTA2.5-Audio"
    preds = tmp_path / "results" / "val" / "preds" / "val@ep=1.0-3.jsonl"
    rows = [json.loads(l) for l in open(preds)]
    assert [r["correct"] for r in rows] == [True, False, True, True] and [r["index"] for r in rows] == [0, 1, 2, 3]
    report = json.load(open(tmp_path / "results" / "val" / "val@ep=1.0-3-report.json"))
    assert report["preds_path"] == str(preds) and all("context" not in r and "audio_context" not in r for r in report["results"])
    # a second evaluation at the same step does not overwrite the first predictions file
    rep2 = DeSTA25Trainer._save_results(me, [dict(prediction="x", label="y")], path, ckpt="ep=1.0-3")
    assert rep2["preds_path"] != rep["preds_path"] and os.path.exists(rep["preds_path"]) and rep2["accuracy_by_sample"] == 0.0
    assert DeSTA25Trainer._save_results(me, [], tmp_path / "results" / "val" / "empty.jsonl")["accuracy_by_sample"] == 0
