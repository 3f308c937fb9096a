class Word2Vec:  # import stub only
    pass
