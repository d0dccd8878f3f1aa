"""Synthetic OpenAlex `works` records for the oa_jsonl tests (test infrastructure): the
fields the reference filter reads (id, title, language, abstract_inverted_index;
reference oa_jsonl.c:371-389) among fields it skips, in varying order and spacing."""
import json

WORDS = ("the of and a in to is for with on we that this by are an as be which from study results using method "
         "data model analysis théorie naïve Schrödinger 量子 \\\"quoted\\\" back\\\\slash tab\\tsep new\\nline "
         "α-decay CO₂ 10.1016/j.x.2020 e.g. i.e. \\u00e9l\\u00e8ve {brace} [bracket] comma,inside colon:inside").split()


def _inv_index(rng, n_words, gaps=False, repeat=False):
    """{"word": [positions]} for an abstract of n_words tokens (raw JSON text, words are already escaped)."""
    toks = [rng.choice(WORDS) for _ in range(n_words)]
    pos = {}
    p = 0
    for t in toks:
        if gaps and rng.random() < 0.15:
            p += rng.randint(1, 3)                 # positions nobody claims
        pos.setdefault(t, []).append(p)
        p += 1
    if repeat and len(toks) > 2:                   # two words claim one position: the later key wins
        a, b = list(pos)[0], list(pos)[-1]
        if a != b:
            pos[b].append(pos[a][0])
    items = list(pos.items())
    rng.shuffle(items)
    sep = rng.choice([",", ", ", " , "])
    return "{" + sep.join('"%s":%s[%s]' % (w, rng.choice(["", " "]), rng.choice([",", ", "]).join(str(i) for i in ps))
                          for w, ps in items) + "}"


def random_record(rng):
    wid = "https://openalex.org/W%d" % rng.randint(10 ** 9, 10 ** 10)
    lang = rng.choice(['"en"'] * 6 + ['"fr"', '"de"', "null", '"zh"'])
    title = rng.choice(["null", '"%s"' % " ".join(rng.choice(WORDS) for _ in range(rng.randint(1, 12)))])
    r = rng.random()
    if r < 0.12:
        abstract = "null"
    elif r < 0.16:
        abstract = "{}"
    else:
        abstract = _inv_index(rng, rng.randint(1, 180), gaps=rng.random() < 0.3, repeat=rng.random() < 0.2)
    other = {
        "doi": "https://doi.org/10.%d/x.%d" % (rng.randint(1000, 9999), rng.randint(1, 10 ** 6)),
        "publication_year": rng.randint(1900, 2026), "cited_by_count": rng.randint(0, 5000),
        "is_retracted": rng.random() < 0.01, "fwci": round(rng.random() * 10, 3), "relevance": -1.5e-3,
        "authorships": [{"author": {"id": "https://openalex.org/A%d" % rng.randint(1, 10 ** 9), "display_name": "N. \"Q\" O'Name"},
                         "institutions": [{"display_name": "Univ. {of} [Brackets]", "country_code": None}]}
                        for _ in range(rng.randint(0, 4))],
        "concepts": [], "biblio": {"volume": None, "issue": "3", "first_page": "1"}, "open_access": {"is_oa": True},
    }
    fields = [('"id"', '"%s"' % wid), ('"title"', title), ('"language"', lang), ('"abstract_inverted_index"', abstract)]
    fields += [('"%s"' % k, json.dumps(v, ensure_ascii=rng.random() < 0.5)) for k, v in other.items()]
    head = fields[:1]                              # id first (as in the OpenAlex dump), the rest shuffled
    rest = fields[1:]
    rng.shuffle(rest)
    colon = rng.choice([":", ": ", " : "])
    comma = rng.choice([",", ", "])
    return "{" + comma.join(k + colon + v for k, v in head + rest) + "}"


def edge_cases():
    """Hand-written records (see SURVEY appendix A.2 for the behaviours they pin)."""
    return [
        '{"id":"W1","title":"A title","language":"en","abstract_inverted_index":{"hello":[0],"world":[1]}}',
        '{"id":"W2","title":null,"language":"en","abstract_inverted_index":{"only":[0],"abstract":[1]}}',
        '{"id":"W3","title":"dropped: french","language":"fr","abstract_inverted_index":{"bonjour":[0]}}',
        '{"id":"W4","title":"dropped: null language","language":null,"abstract_inverted_index":{"x":[0]}}',
        '{"id":"W5","title":"dropped: null abstract","language":"en","abstract_inverted_index":null}',
        '{"id":"W6","title":"dropped: empty abstract","language":"en","abstract_inverted_index":{}}',
        '{"id":"W7","title":"gaps","language":"en","abstract_inverted_index":{"a":[0],"c":[4],"b":[2]}}',
        '{"id":"W8","title":"repeats","language":"en","abstract_inverted_index":{"the":[0,2,4],"cat":[1],"dog":[3],"end":[5]}}',
        '{"id":"W9","title":"esc \\"q\\" \\\\ \\u00e9 é","language":"en","abstract_inverted_index":{"say":[0],"\\"hi\\"":[1],"back\\\\":[2]}}',
        '{ "id" : "W10" , "title" : "spaced" , "language" : "en" , "abstract_inverted_index" : { "a" : [ 0 , 2 ] , "b" : [ 1 ] } }',
        '{"id":"W11","language":"en","abstract_inverted_index":{"no":[0],"title":[1],"key":[2]}}',
        '{"id":"W12","title":"abstract before language","abstract_inverted_index":{"kept":[0]},"language":"en"}',
        '{"id":"W13","title":"late french","abstract_inverted_index":{"late":[0]},"language":"fr"}',
        '{"id":"W14","title":"","language":"en","abstract_inverted_index":{"empty":[0],"title":[1]}}',
        '{"id":"W15","title":"skips","n":-1.5e+3,"t":true,"f":false,"z":null,"s":"str ] } \\" ,","o":{"a":[1,{"b":"}"}]},"l":[[],{}],"language":"en","abstract_inverted_index":{"ok":[0]}}',
        '{"id":"W16","title":"overwrite","language":"en","abstract_inverted_index":{"first":[0],"second":[0],"tail":[1]}}',
        '{"id":"W17","title":"no abstract key","language":"en"}',
        '{"id":"W18","title":"big positions","language":"en","abstract_inverted_index":{"far":[300],"near":[0],"mid":[150]}}',
        '{"id":"W19","title":"tab\\tand unicode 量子","language":"en","abstract_inverted_index":{"量子":[0],"naïve":[1]}}',
        '{"title":"id after","language":"en","abstract_inverted_index":{"x":[0]},"id":"W20"}',
    ]
