"""CPU oracle (test infrastructure only -- see oracle/ref.py header).  Never imported by the
product package ``gluon-nlp_amd/``."""
