"""A scope lint for the Haskell files of tools/ghc_pin that cannot be compiled where they are written (no GHC in the image).

NOT a type checker.  It answers one question a compiler would answer first: is every name the file uses either bound in the file or
brought into scope by one of its import lines -- and by exactly one of them, unqualified (the shim once imported Data.Massiv.Array
wholesale next to Prelude's `zip` and Control.Monad's `forM_`: ambiguous occurrences)?  The tables below say which names each import is
RELIED on to provide (module as of resolver lts-13.16: base 4.12, massiv 0.2.x, massiv-io 0.1.x, kdt 0.2.4, vector 0.12, yaml 0.11,
directory 1.3, filepath 1.4, and the reference's own modules with their export lists, src/*.hs): a name that is used and found in no
table fails the test, so that whoever adds it has to say where it comes from.  Local bindings are collected loosely (any lower-case
identifier in a pattern position anywhere in the file), which can hide a typo inside a function but not a missing import."""
import re

KEYWORDS = set("case of let in where do if then else module import qualified as hiding data type newtype class instance deriving foreign ccall safe unsafe "
               "infixr infixl infix forall _".split())

PRELUDE = set("""map zip zip3 length null not return fmap mapM_ show fromIntegral realToFrac fromEnum toEnum max min maximum minimum take drop splitAt head tail fst snd
    concat replicate error otherwise putStr putStrLn getLine ioError userError flip id const either maybe seq div mod rem quot sum product and or any all
    filter reverse lookup elem words unwords lines unlines read floor round ceiling truncate sqrt pi sin cos exp log abs signum negate subtract
    Int Integer Double Float Bool True False Char String IO Maybe Just Nothing Either Left Right Ordering FilePath Show Eq Ord Num Monad Functor
    mapM sequence sequence_ foldr foldl uncurry curry zipWith concatMap iterate repeat span dropWhile takeWhile print readFile writeFile appendFile""".split())

# what each import line of the kit's Haskell files is relied on to export (only the names the files use need to be here)
EXPORTS = {
    "Foreign": set("""Ptr FunPtr ForeignPtr FinalizerPtr nullPtr plusPtr castPtr castForeignPtr alloca allocaBytes peek poke pokeByteOff peekByteOff withArray withArrayLen
        withMany withForeignPtr newForeignPtr newForeignPtr_ touchForeignPtr mallocForeignPtrArray Storable sizeOf Int32 Int64 Word8 Word32 with new""".split()),
    "Foreign.C.Types": set("CInt CUInt CSize CDouble CChar CLong".split()),
    "Foreign.C.String": set("CString peekCString withCString newCString".split()),
    "Data.ByteString": set("packCStringLen writeFile readFile ByteString take drop null".split()),
    "Control.Exception": set("bracket bracket_ finally".split()),
    "Control.Concurrent": set("runInBoundThread forkIO forkOS".split()),
    "Control.Concurrent.MVar": set("MVar newMVar withMVar takeMVar putMVar".split()),
    "Control.Monad": set("when unless forM forM_ filterM replicateM_ void".split()),
    "Data.IORef": set("IORef newIORef readIORef writeIORef modifyIORef".split()),
    "System.IO.Unsafe": set("unsafePerformIO".split()),
    "Data.KdMap.Static": set("assocs KdMap build inRadius size".split()),
    "Control.DeepSeq": set("deepseq NFData force".split()),
    "Data.List": set("partition sort sortBy".split()),
    # (massiv re-uses Prelude / Control.Monad names: listed so that a wholesale unqualified import shows up as ambiguous)
    "Data.Massiv.Array": set("""U Par Seq Comp Ix2 Array size toList map zip zip3 zipWith unzip forM forM_ mapM mapM_ sum product maximum minimum and or all any
        foldr foldl traverse transpose""".split()) | {":."},
    "Data.Massiv.Array.IO": set("Image".split()),
    "Data.Massiv.Array.Manifest.Vector": set("fromVector toVector".split()),
    "Data.Vector.Storable": set("Vector unsafeFromForeignPtr0".split()),
    "Graphics.ColorSpace": set("Pixel RGB HSI PixelRGB PixelHSI".split()),
    "Linear": set("V3".split()),
    "Data.Yaml": set("decodeFileEither prettyPrintParseException ParseException".split()),
    "System.Directory": set("doesFileExist createDirectoryIfMissing listDirectory".split()),
    "System.FilePath": set("takeBaseName takeExtension".split()) | {"</>", "<.>"},
    "System.IO": set("hFlush stdout IOMode WriteMode hPutStrLn withFile".split()),
    "Data.ByteString.Builder": set("toLazyByteString doubleLE Builder".split()),
    "Data.ByteString.Lazy": set("ByteString writeFile".split()),
    "Data.Serialize": set("getFloat64le runGet Get".split()),
    "Data.Version": set("showVersion".split()),
    "System.Environment": set("getArgs".split()),
    "System.Exit": set("die exitFailure".split()),
    "System.Info": set("arch compilerName compilerVersion os".split()),
    # the reference's own modules (export lists: src/ConfigFile.hs:4-10, src/StarMap.hs:7-10, src/Raytracer.hs:4, src/ImageFilters.hs:5, src/Util.hs:1-2,
    # src/Animation.hs:3-6)
    "Raytracer": set("render writeImg".split()),
    "ImageFilters": set("bloom supersample".split()),
    "Util": set("promptOverwriteFile readSafe normalizePath timeAction padZero".split()),
    "Animation": set("Keyframe camera time Animation scene nFrames interpolation keyframes InterpolationMethod Linear generateFrames validateKeyframes".split()),
    "ConfigFile": set("""Scene safeDistance stepSize bloomStrength bloomDivider starIntensity starSaturation supersampling diskColor diskOpacity diskInner diskOuter resolution
        Camera position lookAt upVec fov Config camera scene""".split()),
    "StarMap": set("Star StarTree StoredStarTree readMapFromFile treeToByteString readTreeFromFile buildStarTree starLookup".split()),
}


# T(..) in an import list brings these constructors / fields with it
WITH_ALL = {"IOMode": {"ReadMode", "WriteMode", "AppendMode"}, "Pixel": {"PixelRGB", "PixelHSI", "PixelRGBA", "PixelY"}, "V3": {"V3"}, "Ix2": {":."}, "Comp": {"Seq", "Par", "ParOn"}}


def strip(src):
    src = re.sub(r"\{-.*?-\}", " ", src, flags=re.S)
    src = re.sub(r'"(?:\\.|[^"\\])*"', '""', src)
    src = re.sub(r"'(?:\\.|[^'\\])'", "' '", src)
    return "\n".join(re.sub(r"(^|\s)--.*$", "", ln) for ln in src.split("\n"))


def imports(src):
    """[(module, qualified alias or None, explicit names or None)]"""
    out = []
    for m in re.finditer(r"^import\s+(qualified\s+)?([A-Z][\w.]*)(?:\s+as\s+(\w+))?(?:\s*\(([^\n]*)\))?\s*$", src, re.M):
        names = None
        if m.group(4) is not None:
            names = set()
            for item in re.findall(r"[A-Za-z_][\w']*(?:\s*\(\.\.\))?|\([^\w\s()]+\)", m.group(4)):
                base = item.replace("(..)", "").strip().strip("()")
                names.add(base)
                if "(..)" in item:
                    names |= WITH_ALL.get(base, set())
        out.append((m.group(2), m.group(3) if m.group(1) else None, names, bool(m.group(1)), m.group(3)))
    return out


def check(path, extra_modules=None):
    """Returns (unknown names, ambiguous names): both must be empty."""
    raw = open(path).read()
    src = strip(raw)
    exports = dict(EXPORTS, **(extra_modules or {}))
    body = "\n".join(ln for ln in src.split("\n") if not re.match(r"^(import|module)\b|^\s+[\w, ]+\) where", ln))
    body = re.sub(r"^module[^\n]*(?:\n\s+[^\n]*)*?where", "", body, count=1, flags=re.M)
    # names bound in the file: top-level definitions / signatures, and any lower-case identifier in a pattern position
    bound = set(re.findall(r"^([a-z_][\w']*)\s*::", body, re.M)) | set(re.findall(r"^([a-z_][\w']*)\b[^=\n]*=", body, re.M))
    bound |= set(re.findall(r"^(?:data|newtype|type)\s+([A-Z]\w*)", body, re.M))
    bound |= set(re.findall(r'^foreign import ccall(?:\s+(?:safe|unsafe))?\s*(?:"[^"]*")?\s+([a-z_][\w\']*)', body, re.M))
    for m in re.finditer(r"^data\s+\w+\s*=\s*(\w+)\s*\{([^}]*)\}", body, re.M):     # record constructor + fields
        bound.add(m.group(1))
        bound |= set(re.findall(r"([a-z_][\w']*)\s*::", m.group(2)))
    EQ = r"(?<![/<>=!:|&*+.-])=(?![=>])"                                 # a binding's `=`, not part of an operator
    pats = re.findall(r"\\([^\n\\]*?)->", body)                            # lambda arguments
    pats += re.findall(r"^\s*([^\n=$\\]*?)<-", body, re.M)                  # do / comprehension binds at the start of a statement
    pats += re.findall(r"\|\s*([^\n$\\]*?)<-", body) + re.findall(r",\s*\(?([a-z_][\w', ]*)\)?\s*<-", body)   # comprehension generators
    pats += re.findall(r"\blet\s+([^\n=$\\]*?)" + EQ, body)               # let pat =
    pats += re.findall(r"^\s+([^\n=$\\|]*?)" + EQ, body, re.M)              # continuation lines of a let / where block: pat =
    pats += re.findall(r"^([a-z_][\w']*[^\n=$\\|]*?)" + EQ, body, re.M)     # top-level equations: name args =
    pats += re.findall(r"^([a-z_][\w']*[^\n=$\\|]*)\n\s+\|", body, re.M)      # top-level equations with guards on the following lines
    pats += re.findall(r";\s*([^\n=;$\\]*?)" + EQ, body)                   # let a = ..; b = ..
    pats += re.findall(r"^\s+([A-Za-z_(\[][^\n$\\<=]*?)\s*->", body, re.M)  # case alternatives: pat ->
    for pat in pats:
        bound |= set(re.findall(r"(?<![\w.'])([a-z_][\w']*)", pat))
    unq, qual = {}, {}
    for mod, alias, names, is_q, as_ in imports(src):
        have = exports.get(mod)
        assert have is not None, f"{path}: no export table for module {mod}"
        if names is not None:
            have = have | set().union(*(WITH_ALL.get(n, set()) for n in names))
            missing = names - have
            assert not missing, f"{path}: import {mod} ({', '.join(sorted(missing))}): not in the export table of {mod}"
            have = names
        if is_q:
            qual.setdefault(alias or mod, set()).update(have)
        else:
            for n in have:
                unq.setdefault(n, set()).add(mod)
            if as_:
                qual.setdefault(as_, set()).update(have)
    unknown, ambiguous = set(), set()
    for q, n in re.findall(r"(?<![\w'])([A-Z]\w*)\.([a-z_A-Z][\w']*|[:!<>/.*+|&=-]+)", body):
        if q in qual:
            if n not in qual[q]:
                unknown.add(f"{q}.{n}")
    plain = re.sub(r"(?<![\w'])[A-Z]\w*\.(?:[a-zA-Z_][\w']*|[:!<>/.*+|&=-]+)", " ", body)
    for n in set(re.findall(r"(?<![\w.'])([A-Za-z_][\w']*)", plain)):
        if n in KEYWORDS or n in bound:
            continue
        srcs = unq.get(n, set()) | ({"Prelude"} if n in PRELUDE else set())
        if not srcs:
            unknown.add(n)
        elif len(srcs) > 1:
            ambiguous.add(f"{n} ({', '.join(sorted(srcs))})")
    for op in ("</>", "<.>"):
        if op in plain and op not in unq:
            unknown.add(op)
    return unknown, ambiguous
