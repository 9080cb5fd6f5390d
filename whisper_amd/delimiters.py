"""Delimiter string factories with `.regex` attributes (mirror of urro_whisper/delimiters.py:3-29).
The reference looks closing brackets up in python-bidi's MIRRORED table; the common pairs are listed here."""
_MIRRORED = {"[": "]", "(": ")", "{": "}", "<": ">", "«": "»", "‹": "›", "「": "」", "『": "』", "【": "】",
             "〈": "〉", "《": "》", "〔": "〕", "⟨": "⟩", "⟦": "⟧", "]": "[", ")": "(", "}": "{", ">": "<"}


def HYPHEN(count: int = 1) -> str:
    return " " + "-" * count


HYPHEN.regex = r"(\s)?\-\s"


def GREATER_THAN(count: int = 1) -> str:
    return " " + ">" * count


GREATER_THAN.regex = r"(\s)?\>\s"


def SPEAKER(ch: str = "[", short: bool = False) -> str:
    return " " + ch + ("S1" if short else "SPEAKER 1") + _MIRRORED.get(ch, ch)


SPEAKER.regex = r"(\s)?\[SPEAKER\s\d\]\s"


def PERSON(ch: str = "[", short: bool = False) -> str:
    return " " + ch + ("P1" if short else "PERSON 1") + _MIRRORED.get(ch, ch)


PERSON.regex = r"(\s)?\[PERSON\s\d\]\s"
