"""Vocabulary data for the bag-of-words tests: a parser of the DBoW2 / OpenCV-FileStorage YAML the reference ships
(demo/voc.yml.gz, data) and a seeded generator of synthetic vocabularies of the same shape, written in the same
format so that the product's loader is exercised on it.  Test infrastructure."""
import gzip
import os
import re

import numpy as np

REF_VOC = "/root/reference/demo/voc.yml.gz"          # exists only in the build container (never read by -m gpu tests)

_NODE = re.compile(rb'nodeId:\s*(\d+),\s*parentId:\s*(\d+),\s*weight:\s*([-+0-9.eE]+),\s*descriptor:\s*"([01]*)"', re.S)
_WORD = re.compile(rb'wordId:\s*(\d+),\s*nodeId:\s*(\d+)')


def bits_to_bytes(s):
    """DBoW2 BRIEF string -> 32 bytes.  The string is boost::dynamic_bitset MSB-first text: character c holds bit
    (len-1-c); bit m + 8k of the bitset is bit m of descriptor byte k (BOW.h:95-107)."""
    b = np.frombuffer(s, np.uint8) - 48
    assert b.size == 256
    return np.packbits(b[::-1], bitorder="little")


def bytes_to_bits(d):
    return bytes(np.unpackbits(np.asarray(d, np.uint8), bitorder="little")[::-1] + 48)


def parse(path):
    """dict(k, L, scoring, weighting, node_id, parent_id, weight, desc[n,32], word_node) in file order."""
    raw = gzip.open(path, "rb").read() if path.endswith(".gz") else open(path, "rb").read()
    head = raw[:400]
    g = lambda key: int(re.search(rb"\n\s+" + key + rb":\s*(\d+)", head).group(1))
    iw = raw.index(b"words:")
    nodes = _NODE.findall(raw[:iw])
    words = _WORD.findall(raw[iw:])
    n = len(nodes)
    out = dict(k=g(rb"k"), L=g(rb"L"), scoring=g(rb"scoringType"), weighting=g(rb"weightingType"),
               node_id=np.array([int(t[0]) for t in nodes], np.int32), parent_id=np.array([int(t[1]) for t in nodes], np.int32),
               weight=np.array([float(t[2].rstrip(b".") if t[2].endswith(b".") else t[2]) for t in nodes], np.float64),
               desc=np.zeros((n, 32), np.uint8))
    for i, t in enumerate(nodes):
        out["desc"][i] = bits_to_bytes(t[3])
    wn = np.zeros(len(words), np.int32)
    for w, nid in words:
        wn[int(w)] = int(nid)
    out["word_node"] = wn
    return out


def synth(k=8, L=3, seed=7, flip=24, ragged=False):
    """Seeded synthetic vocabulary, complete k-ary tree of depth L, saved in DBoW2's order (a parent's children
    together, depth first from a stack as TemplatedVocabulary::save walks it).  Children are the parent's
    descriptor with `flip` random bits toggled so that descents are decisive but ties do occur; idf-like weights."""
    rng = np.random.default_rng(seed)
    node_id, parent_id, weight, desc, word_node = [], [], [], [], []
    descs = {0: rng.integers(0, 256, 32, dtype=np.uint8)}
    depth = {0: 0}
    next_id = 1
    stack = [0]
    while stack:
        p = stack.pop()
        kids = []
        nk = k if not ragged or rng.random() < 0.7 else int(rng.integers(2, k + 1))   # the reference's tree has 2..8 children per node
        for _ in range(nk):
            nid = next_id
            next_id += 1
            bits = np.unpackbits(descs[p])
            idx = rng.choice(256, flip, replace=False)
            bits[idx] ^= 1
            descs[nid] = np.packbits(bits)
            depth[nid] = depth[p] + 1
            leaf = depth[nid] == L or (ragged and depth[nid] >= 1 and rng.random() < 0.1)    # ... and leaves above level L
            node_id.append(nid); parent_id.append(p)
            weight.append(float(np.round(rng.uniform(0.5, 9.0), 6)) if leaf else 0.0)
            desc.append(descs[nid])
            if leaf:
                word_node.append(nid)
            else:
                kids.append(nid)
        stack.extend(kids)
    # a few zero-weight words (stop words): DBoW2 skips their occurrences
    weight = np.array(weight)
    leaves = np.flatnonzero(weight > 0)
    weight[rng.choice(leaves, max(1, len(leaves) // 50), replace=False)] = 0.0
    return dict(k=k, L=L, scoring=0, weighting=0, node_id=np.array(node_id, np.int32), parent_id=np.array(parent_id, np.int32),
                weight=weight, desc=np.array(desc, np.uint8), word_node=np.array(word_node, np.int32))


def train(images_desc, k=8, L=3, seed=1, iters=4, min_leaf=6):
    """Hierarchical k-medians on real descriptors (the way DBoW2 builds a vocabulary: Hamming distance, majority-bit
    centres), idf weights log(N / N_w) over the training images.  images_desc: list of [n_i, 32] uint8 arrays.  The tree
    is ragged where clusters run empty or small; saved in DBoW2's order."""
    rng = np.random.default_rng(seed)
    img_of = np.concatenate([np.full(len(d), i, np.int32) for i, d in enumerate(images_desc)])
    bits = np.unpackbits(np.concatenate(images_desc), axis=1).astype(np.int32)

    def assign(idx, centres):
        b = bits[idx]
        return np.argmin(b @ (1 - centres).T + (1 - b) @ centres.T, axis=1)

    node_id, parent_id, weight, desc, word_node = [], [], [], [], []
    members = {0: np.arange(len(bits))}
    depth = {0: 0}
    next_id = 1
    stack = [0]
    while stack:
        p = stack.pop()
        idx = members.pop(p)
        kk = min(k, len(idx))
        centres = bits[rng.choice(idx, kk, replace=False)].copy()
        for _ in range(iters):
            a = assign(idx, centres)
            for c in range(kk):
                m = idx[a == c]
                if len(m):
                    centres[c] = (2 * bits[m].sum(0) >= len(m)).astype(np.int32)
        a = assign(idx, centres)
        kids = []
        for c in range(kk):
            m = idx[a == c]
            if len(m) == 0:
                continue
            nid = next_id
            next_id += 1
            depth[nid] = depth[p] + 1
            leaf = depth[nid] == L or len(m) < min_leaf
            node_id.append(nid); parent_id.append(p)
            desc.append(np.packbits(centres[c].astype(np.uint8)))
            if leaf:
                n_img = len(np.unique(img_of[m]))
                weight.append(float(np.log(len(images_desc) / n_img)))
                word_node.append(nid)
            else:
                weight.append(0.0)
                members[nid] = m
                kids.append(nid)
        stack.extend(kids)
    return dict(k=k, L=L, scoring=0, weighting=0, node_id=np.array(node_id, np.int32), parent_id=np.array(parent_id, np.int32),
                weight=np.array(weight), desc=np.array(desc, np.uint8), word_node=np.array(word_node, np.int32))


def write_yaml(v, path):
    """Write `v` in the DBoW2 YAML layout (gzip when the name ends in .gz)."""
    lines = ["%YAML:1.0", "vocabulary:", "   k: %d" % v["k"], "   L: %d" % v["L"], "   scoringType: %d" % v["scoring"],
             "   weightingType: %d" % v["weighting"], "   nodes:"]
    for i in range(len(v["node_id"])):
        w = repr(float(v["weight"][i]))
        if w.endswith(".0"):
            w = w[:-1]                                  # OpenCV writes "0." / "3."
        lines.append("      - { nodeId:%d, parentId:%d, weight:%s," % (v["node_id"][i], v["parent_id"][i], w))
        lines.append('          descriptor:"%s" }' % bytes_to_bits(v["desc"][i]).decode())
    lines.append("   words:")
    for w, nid in enumerate(v["word_node"]):
        lines.append("      - { wordId:%d, nodeId:%d }" % (w, nid))
    data = ("\n".join(lines) + "\n").encode()
    if path.endswith(".gz"):
        with gzip.open(path, "wb") as f:
            f.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data)


def keyframe_descriptors(v, rng, n=300, noise=10):
    """Descriptors near random leaves of the vocabulary (so that words repeat and descents are realistic)."""
    leaves = np.flatnonzero(np.isin(v["node_id"], v["word_node"]))
    pick = rng.choice(leaves, n)
    d = v["desc"][pick].copy()
    bits = np.unpackbits(d, axis=1)
    for r in range(n):
        bits[r, rng.choice(256, noise, replace=False)] ^= 1
    return np.packbits(bits, axis=1)


def keyframe_sequence(v, rng, n_kf=10, n=150, keep=0.7, noise=10):
    """Descriptor sets of consecutive keyframes: each keeps ~`keep` of the previous one's features (re-observed with a few
    flipped bits) and replaces the rest, so that neighbours score high and distant keyframes low."""
    out = [keyframe_descriptors(v, rng, n, noise)]
    for _ in range(1, n_kf):
        d = out[-1].copy()
        bits = np.unpackbits(d, axis=1)
        for r in range(n):
            bits[r, rng.choice(256, 3, replace=False)] ^= 1
        d = np.packbits(bits, axis=1)
        repl = rng.random(n) > keep
        d[repl] = keyframe_descriptors(v, rng, int(repl.sum()), noise)
        out.append(d)
    return out


def have_reference_vocabulary():
    return os.path.exists(REF_VOC)
