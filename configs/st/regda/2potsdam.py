"""st.regda.2potsdam: Vaihingen -> Potsdam (the names of the reference's configs/st/regda/2potsdam.py)."""
from configs.st.regda._surface import install

install(globals(), 'potsdam')
