"""Per-hit annotation lines of the reference's recorded runs (tutorial/*.out), parsed into fields, and the same fields
formatted from a domain record the way p7_tophits_Domains prints them (p7_tophits.c:1322-1378)."""
import math

import oracle_lib as ol

RUNS = [  # output file, model file, target FASTA, --fs
    ("PTH2.out", "PTH2.bhmm", "target-PTH2.fa", False),
    ("AMP_N.out", "AMP_N.bhmm", "target-AMP_N.fa", False),
    ("MET-ct4.out", "MET-ct4.bhmm", "target-MET.fa", False),
    ("AMP_N-fs.out", "AMP_N.bhmm", "target-AMP_N.fa", True),
]


def annotation_lines(outfile):
    """[[fields of every ' ! ...' line] per query], brackets ('..', '[.', ...) dropped."""
    per_query, cur = [], None
    for line in open(ol.GOLDEN + "/" + outfile):
        if line.startswith("Query:"):
            cur = []
            per_query.append(cur)
        elif line.startswith(" ! ") or line.startswith(" ? "):
            cur.append([t for t in line.split()[1:] if not set(t) <= set("[].")])
    return per_query


def fields_of(d, sq_len, fs, with_env):
    """score bias Evalue hmm-from hmm-to ali-from ali-to [env-from env-to | shifts stops] sq-len acc, as printed."""
    acc = d.oasc / (1.0 + abs(float(d.jenv - d.ienv) / 3))
    out = ["%.1f" % d.bitscore, "%.1f" % (d.dombias * 1.44269504088896341), "%.2g" % math.exp(d.lnP), str(d.ihmm), str(d.jhmm), str(d.iali), str(d.jali)]
    if fs:
        out += [str(d.n_shifted_codons), str(d.n_stops)]
    elif with_env:
        out += [str(d.ienv), str(d.jenv)]
    return out + [str(sq_len), "%.2f" % acc]


def targets_blocks(outfile):
    """The 'Scores for complete hits' block of every query in a recorded bathsearch output, up to its last hit line."""
    blocks = []
    lines = open(ol.GOLDEN + "/" + outfile).read().split("\n")
    i = 0
    while i < len(lines):
        if lines[i].startswith("Scores for complete hits"):
            j = i + 3
            while j < len(lines) and lines[j].strip():
                j += 1
            blocks.append("\n".join(lines[i:j]) + "\n")
            i = j
        i += 1
    return blocks


def annotation_heads(outfile):
    """For every hit of a recorded output: '>> name  desc', the two header lines and the hit's line."""
    lines = open(ol.GOLDEN + "/" + outfile).read().split("\n")
    return ["\n".join(lines[i:i + 4]) + "\n" for i in range(len(lines)) if lines[i].startswith(">> ")]


def statistics_blocks(outfile):
    """'Internal pipeline statistics summary' of every query, without the two timing lines."""
    lines = open(ol.GOLDEN + "/" + outfile).read().split("\n")
    return ["\n".join(lines[i:i + 9]) + "\n" for i in range(len(lines)) if lines[i].startswith("Internal pipeline statistics summary")]
