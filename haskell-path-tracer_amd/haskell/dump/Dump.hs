{-# LANGUAGE CPP #-}
{-# LANGUAGE FlexibleInstances #-}
{-# LANGUAGE ScopedTypeVariables #-}
{-# LANGUAGE TypeApplications #-}
{-# LANGUAGE TypeOperators #-}

-- | @ptmi-dump@: the GHC-side half of the parity experiment of the MI355X build (DESIGN.md section 2).
--
-- The MI355X build of this path (libptmi) equals its own C restatement of @render@ bit for bit, but that restatement is pinned
-- against THIS repository only where this repository holds tests (the eight intersection properties).  Everything downstream
-- of the RNG rests on seven named assumptions, A1-A7, that only a real run of the reference can settle.  This program is that
-- run: it uses nothing beyond the packages the reference already depends on, renders with the reference's own @render@ on
-- Accelerate's CPU backend from INJECTED seeds, and writes what it sees to one flat binary file.  On the other side
-- @tools/compare_ghc_dump.py@ (MI355X repository) renders the same inputs through the oracle (and through libptmi when a GPU is
-- present) and says, assumption by assumption, what holds.
--
-- STATUS: SOURCE ONLY, NEVER COMPILED -- the build container of the MI355X repository has no GHC.  Where a guess about an
-- un-vendored dependency was needed it is marked GUESS below; each is one line to adjust.
--
-- Build and run (in a checkout of the reference with this file as @dump/Dump.hs@ and @dump/tracer.cabal.diff@ applied):
--
-- > cabal run ptmi-dump -- ptmi_dump.bin          # ~70 MB: 800x600, the reference's own size and iteration limit
--
-- What is written, and which assumption it settles (A1-A7 as in DESIGN.md section 2 of the MI355X repository):
--
--   * @words@         the three seed words per pixel handed to 'createWith' (a fixed function of the pixel index)
--   * @created@       @run (createWith (use words))@, every plane of its representation in 'toVectors' order
--                       -> A2 (createWith = PractRand's three-word seeding) and A3 (which plane holds which state word)
--   * @probe_words@   four successive raw @random \@Word32@ draws from each of the first 64 states     -> A1 (the raw step)
--   * @probe_floats@  four successive @random \@Float@ draws from the same states                      -> A4 (word -> float)
--   * @inline_1/2@    the seven planes after one and two calls of @runN (render Inline) screenPixels@   -> A6, A7 and the whole path
--   * @streams_1/2@   the same through @render Streams@                                                 -> A5 (which seed survives combine)
--
-- File format, little endian: @"PTMIDUMP"@, u32 version (1), u32 width, u32 height, u32 iteration limit, u32 sections; then per
-- section a 16-byte zero-padded name, u32 planes, and per plane u32 bytes per element, u64 elements, the elements.
module Main where

import qualified Data.Array.Accelerate         as A
import           Data.Array.Accelerate          ( Acc
                                                , Exp
                                                , Matrix
                                                , Vector
                                                , Z(..)
                                                , (:.)(..)
                                                )
import           Data.Array.Accelerate.IO.Data.Vector.Storable
                                                ( toVectors )
import           Data.Array.Accelerate.Linear   ( )                     -- Elt (V3 Float)
import           Data.Array.Accelerate.LLVM.Native
                                                ( run
                                                , runN
                                                )
import           Data.Array.Accelerate.System.Random.SFC
                                                ( Random
                                                , SFC32
                                                , createWith
                                                , random
                                                , runRandom
                                                )
import           Data.Bits
import qualified Data.ByteString               as BS
import qualified Data.ByteString.Builder       as B
import qualified Data.ByteString.Char8         as BC
import qualified Data.Vector.Storable          as V
import           Data.Word
import           Foreign.Ptr                    ( castPtr )
import           Foreign.Storable               ( Storable
                                                , sizeOf
                                                )
import           Linear                         ( V3(..) )
import           System.Environment             ( getArgs )
import           System.IO

import           Scene.Objects                  ( RenderResult )
import           Scene.Trace                    ( Algorithm(..)
                                                , render
                                                )
import           Scene.World                    ( initialCamera )
import           Util                           ( scalar
                                                , screenHeight
                                                , screenPixels
                                                , screenShape
                                                , screenWidth
                                                )

-- ---------------------------------------------------------------------------------------------------------------------
-- the injected seeds
-- ---------------------------------------------------------------------------------------------------------------------

-- | murmur3's 32-bit finaliser: a fixed, well-mixed word per (pixel, k).  The words are written to the dump, so the reader does
-- not have to know this function.
mix32 :: Word32 -> Word32
mix32 h0 = h5
 where
  h1 = h0 `xor` (h0 `shiftR` 16)
  h2 = h1 * 0x85ebca6b
  h3 = h2 `xor` (h2 `shiftR` 13)
  h4 = h3 * 0xc2b2ae35
  h5 = h4 `xor` (h4 `shiftR` 16)

-- | What 'Util.genSeeds' draws from the operating system's entropy (src/Util.hs:122-127), as a fixed table.
-- GUESS: the element type 'createWith' takes is a triple of 'Word32' (mwc-random's 'uniform' fills whatever it is at
-- src/Util.hs:125-127; for the 64-bit generator of sfc-random-accelerate it is a triple of Word64).
seedWords :: Matrix (Word32, Word32, Word32)
seedWords = A.fromFunction screenShape $ \(Z :. y :. x) ->
  let i = fromIntegral (y * fromIntegral screenWidth + x) :: Word32
      w k = mix32 ((3 * i + k) `xor` 0x5EED1234)
  in  (w 0, w 1, w 2)

-- | @initialOutput@ (src/Util.hs:204-205) with the table instead of entropy.
initialFrom :: Acc (Matrix SFC32) -> Acc RenderResult
initialFrom = A.map (A.T2 (A.constant (V3 0.0 0.0 0.0 :: V3 Float)))

-- ---------------------------------------------------------------------------------------------------------------------
-- probes of the generator
-- ---------------------------------------------------------------------------------------------------------------------

probeCount :: Int
probeCount = 64

probeStates :: Acc (Matrix SFC32) -> Acc (Vector SFC32)
probeStates = A.take (A.constant probeCount) . A.flatten

-- | Four successive draws of one type from every probed state (the state threads left to right through the applicative,
-- exactly as in 'Util.genVec', src/Util.hs:114-118).
draw4 :: forall a . A.Elt a => Random (Exp SFC32) (Exp a) -> Acc (Vector SFC32) -> Acc (Vector (a, a, a, a))
draw4 gen = A.map $ \g -> let ((a, b, c, d), _) = runRandom g ((,,,) <$> gen <*> gen <*> gen <*> gen) in A.T4 a b c d

-- ---------------------------------------------------------------------------------------------------------------------
-- flat planes out of Accelerate's nested representation
-- ---------------------------------------------------------------------------------------------------------------------

-- | 'toVectors' returns nested pairs of storable vectors, one per scalar leaf of the element type (app/Main.hs:350 shows the
-- nesting of the colour part).  The leaves are written in nesting order WITHOUT assuming the shape of the nesting: how SFC32 is
-- represented is exactly assumption A3.
class Planes v where
  planes :: v -> IO [(Int, Int, BS.ByteString)]           -- bytes per element, elements, the elements

instance Planes () where
  planes () = return []

instance (Planes a, Planes b) => Planes (a, b) where
  planes (a, b) = (++) <$> planes a <*> planes b

instance Storable e => Planes (V.Vector e) where
  planes v = do
    let size = sizeOf (undefined :: e)
    bytes <- V.unsafeWith v $ \p -> BS.packCStringLen (castPtr p, V.length v * size)
    return [(size, V.length v, bytes)]

section :: Planes v => String -> v -> IO B.Builder
section name v = do
  ps <- planes v
  let name16 = BS.take 16 (BC.pack name `BS.append` BS.replicate 16 0)
      plane (size, n, bytes) = B.word32LE (fromIntegral size) <> B.word64LE (fromIntegral n) <> B.byteString bytes
  return $ B.byteString name16 <> B.word32LE (fromIntegral $ length ps) <> mconcat (map plane ps)

-- ---------------------------------------------------------------------------------------------------------------------

main :: IO ()
main = do
  args <- getArgs
  let path = case args of
        (p : _) -> p
        []      -> "ptmi_dump.bin"

  let created = run (createWith (A.use seedWords)) :: Matrix SFC32
      probed  = run (probeStates (A.use created))
      rawWords  = run (draw4 (random :: Random (Exp SFC32) (Exp Word32)) (A.use probed))
      rawFloats = run (draw4 (random :: Random (Exp SFC32) (Exp Float)) (A.use probed))
      start   = run (initialFrom (A.use created))
      -- compileFor (app/Main.hs:188-191): one call = one sample
      step algorithm = runN (render algorithm) screenPixels (scalar initialCamera)
      inline1  = step Inline start
      inline2  = step Inline inline1
      streams1 = step Streams start
      streams2 = step Streams streams1

  sections <- sequence
    [ section "words"        (toVectors seedWords)
    , section "created"      (toVectors created)
    , section "probe_states" (toVectors probed)
    , section "probe_words"  (toVectors rawWords)
    , section "probe_floats" (toVectors rawFloats)
    , section "inline_1"     (toVectors inline1)
    , section "inline_2"     (toVectors inline2)
    , section "streams_1"    (toVectors streams1)
    , section "streams_2"    (toVectors streams2)
    ]
  let header = B.byteString (BC.pack "PTMIDUMP") <> B.word32LE 1 <> B.word32LE (fromIntegral screenWidth)
        <> B.word32LE (fromIntegral screenHeight) <> B.word32LE 15                 -- `traceInline 15` (src/Scene/Trace.hs:200)
        <> B.word32LE (fromIntegral $ length sections)
  withBinaryFile path WriteMode $ \h -> B.hPutBuilder h (header <> mconcat sections)
  putStrLn $ "wrote " ++ path ++ "; compare with: python tools/compare_ghc_dump.py " ++ path
