"""`-m "not gpu"`: the lead-byte logic of `utf8_transcode_kernel` (hip/utf8.hip) restated chunk by chunk in Python.

The kernel decodes 64 bytes per step.  Which bytes are LEADS is a chain - byte p is a lead iff some lead q < p has
q + length(byte q) == p, lengths from the lead byte alone (`sz_rune_decode_unchecked`) - that the kernel follows with six rounds
of pointer doubling per chunk, or, since round 4, SKIPS when the chunk is well formed: every byte that is not a continuation byte
is followed by exactly the continuation bytes it announces (looked up in this chunk and the next one), and the chunk's head holds
exactly the bytes the previous chunk left hanging.  This file checks, on clean and on damaged text, that the shortcut is taken
only where it yields the chain's own leads, and that chunking with the `hanging` count carries the chain across chunk borders."""
import random

import pytest


def length_of(byte):
    return 1 + (byte >= 0xC0) + (byte >= 0xE0) + (byte >= 0xF0)


def leads_by_the_chain(data):
    leads, at = [], 0
    while at < len(data):
        leads.append(at)
        at += length_of(data[at])
    return leads


def leads_by_chunks(data):
    """The kernel's arithmetic: per 64-byte chunk, the shortcut when its test passes, else the chain from `hanging`.  Returns the
    leads and how many chunks took the shortcut."""
    leads, shortcuts, hanging = [], 0, 0
    for base in range(0, len(data), 64):
        chunk, ahead = data[base:base + 64], data[base + 64:base + 128]
        valid = len(chunk)
        continuing = [(b & 0xC0) == 0x80 for b in chunk] + [False] * (64 - valid)
        continuing_ahead = [(b & 0xC0) == 0x80 for b in ahead] + [False] * (64 - len(ahead))
        following = lambda lane: [(continuing + continuing_ahead)[lane + 1 + k] for k in range(4)]  # noqa: E731

        def announced(lane):
            sequence = length_of(chunk[lane])
            window = following(lane)
            return all(window[k] for k in range(sequence - 1)) and not window[sequence - 1]

        head_as_left = hanging < 4 and all(continuing[k] for k in range(hanging)) and not continuing[hanging] if hanging < 64 else False
        well_formed = head_as_left and all(announced(lane) for lane in range(valid) if not continuing[lane])
        high = any(b >= 0x80 for b in chunk)
        if not high and not hanging:
            here = list(range(valid))
        elif well_formed:
            here, shortcuts = [lane for lane in range(valid) if not continuing[lane]], shortcuts + 1
        else:
            here, at = [], hanging
            while at < valid:
                here.append(at)
                at += length_of(chunk[at])
        leads += [base + lane for lane in here]
        if here:
            end = here[-1] + length_of(chunk[here[-1]])
            hanging = end - 64 if end > 64 else 0
        else:
            hanging = hanging - 64 if hanging >= 64 else 0
    return leads, shortcuts


ALPHABET = ["a", "b", " ", "é", "ß", "中", "文", "𝄞", "😀"]


@pytest.mark.parametrize("seed", range(40))
def test_chunked_leads_are_the_chains_leads(seed):
    rng = random.Random(seed)
    text = "".join(rng.choice(ALPHABET) for _ in range(rng.randint(0, 700)))
    raw = bytearray(("x" * (seed % 5) + text).encode())
    if seed % 3 == 1 and raw:  # damage: strays, leads without tails, tails without leads
        for _ in range(rng.randint(1, 8)):
            raw[rng.randrange(len(raw))] = rng.choice([0x80, 0xBF, 0xC3, 0xE4, 0xF0, 0x41, 0xFF])
    if seed % 3 == 2 and raw:
        raw = raw[: len(raw) - rng.randint(0, 3)] + bytes([rng.choice([0xC3, 0xE4, 0xF0])])  # a lead as the last byte
    data = bytes(raw)
    leads, shortcuts = leads_by_chunks(data)
    assert leads == leads_by_the_chain(data), (seed, len(data))
    if seed % 3 == 0 and len(data) > 128 and any(b >= 0x80 for b in data):
        assert shortcuts > 0, "clean multibyte text is expected to take the shortcut"


def test_pure_noise():
    rng = random.Random(99)
    for _ in range(200):
        data = bytes(rng.randrange(256) for _ in range(rng.randint(0, 300)))
        assert leads_by_chunks(data)[0] == leads_by_the_chain(data)
