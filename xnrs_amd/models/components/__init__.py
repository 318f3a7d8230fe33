from .news_encoding import TextEncoder
from .user_encoding import UserEncoder
from .parent import ParentRec
from . import layers, scoring
