"""st.regda.2vaihingen: Potsdam -> Vaihingen (the names of the reference's configs/st/regda/2vaihingen.py)."""
from configs.st.regda._surface import install

install(globals(), 'vaihingen')
